"""B=1 graph-captured rollout alone (for rocprofv3 --kernel-trace --stats): where one 1.3 ms step goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transformerbasednavierstokesolver_amd import harness, ops, synth

eng = ops.resolve_engine(sys.argv[1] if len(sys.argv) > 1 else None)
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = synth.NS_CONFIG
sd = synth.synth_state_dict(cfg, seed=0)
dev = torch.device("cuda:0")
model = harness.build_model(cfg, sd, dev, engine=eng).eval()
pos, a, u = synth.ns_batch(max(bsz, 1), seed=100)
x, fx = torch.from_numpy(pos).to(dev), torch.from_numpy(a).to(dev)
gr = harness.GraphedRollout(model, x[:bsz], fx[:bsz])
gr.run(fx[:bsz], 2)
ts = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gr.run(fx[:bsz], 20)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"B={bsz} engine {eng}: median {20/ts[3]:.1f} steps/s  best {20/ts[0]:.1f}  ({1e6*ts[3]/20:.0f} us/step)")
