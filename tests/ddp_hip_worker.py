"""Rank body of tests/test_gpu_ddp.py (started by `python -m torch.distributed.run`, one process per rank).
Runs `harness.train_step` of the real HIP model under a process group: FusedAdamW + its flat gradient bucket,
all-reduce(SUM) through `opt.sync`.  PA2D_DIST_BACKEND=gloo lets two ranks share the single GPU of a test box;
on a multi-GPU node the same body runs over "nccl" (= RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from transformerbasednavierstokesolver_amd import synth, harness, ddp  # noqa: E402
from transformerbasednavierstokesolver_amd.optim import FusedAdamW  # noqa: E402
from transformerbasednavierstokesolver_amd.utils.testloss import FusedTestLoss  # noqa: E402


def run(rank, world, dev, nsteps, global_batch, out):
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    sd = synth.synth_state_dict(cfg, seed=61)
    model = harness.build_model(cfg, sd, dev).train()
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=10)
    pos, a, u = synth.ns_batch(global_batch, seed=62)
    x, fx, yy = (torch.from_numpy(t).to(dev) for t in (pos, a, u[..., :3]))
    x, fx, yy = ddp.shard_batch((x, fx, yy), rank, world)
    losses = []
    for _ in range(nsteps):
        loss, _ = harness.train_step(model, opt, sched, x, fx, yy, grad_sync=opt.sync,
                                     loss_fn=FusedTestLoss(size_average=False))
        losses.append(float(loss))
    if rank == 0:
        np.savez(out, losses=np.asarray(losses), world=world, nbytes=opt.sync.nbytes,
                 placeholder_grad_none=model.placeholder.grad is None,
                 **{"p." + k: v.detach().cpu().numpy() for k, v in model.named_parameters()})


def main():
    out, nsteps, global_batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PA2D_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = local if backend == "nccl" else local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    run(rank, world, dev, nsteps, global_batch, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
