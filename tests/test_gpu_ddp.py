"""N>1 on the PRODUCT model: two ranks run `harness.train_step` of the HIP Transolver with
FusedAdamW + the flat gradient bucket under a process group and must land on the parameters of the
single-process step at the global batch.  The test box has one GPU, so both ranks share cuda:0 and the
collective runs over gloo (PA2D_DIST_BACKEND=gloo); the rank body (tests/ddp_hip_worker.py) is the one a
multi-GPU node runs over RCCL.  Also: `python bench.py --gpus 2` must start its own ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import rel_l2, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    return dict(os.environ, PA2D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")


def test_two_ranks_of_the_hip_model_equal_the_global_batch_step(tmp_path):
    out2, out1 = str(tmp_path / "w2.npz"), str(tmp_path / "w1.npz")
    worker = os.path.join(ROOT, "tests", "ddp_hip_worker.py")
    nsteps, gb = 2, 4
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, out2, str(nsteps),
                        str(gb)], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, worker, out1, str(nsteps), str(gb)], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    two, one = np.load(out2), np.load(out1)
    assert int(two["world"]) == 2 and int(one["world"]) == 1
    assert bool(two["placeholder_grad_none"]) and bool(one["placeholder_grad_none"])
    # rank 0 of the 2-rank run logs its SHARD's loss; the parameters carry the global-batch update
    worst = max((rel_l2(two[k], one[k]), k) for k in one.files if k.startswith("p."))
    assert worst[0] <= 2e-5, worst
    assert np.array_equal(two["p.placeholder"], one["p.placeholder"])      # never stepped, never decayed


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks itself."""
    env = _env()
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--batch-per-gpu", "2", "--no-rollout", "--no-folded-leg", "--no-split-leg", "--no-darcy-leg"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dist_backend"] == "gloo" and rec["rccl_ranks"] == 0
    assert rec["config"]["global_batch"] == 4 and rec["value"] > 0
