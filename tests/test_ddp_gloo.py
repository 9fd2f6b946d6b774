"""CPU, world_size 2 over gloo: the N>1 data-parallel plumbing (batch sharding + flat gradient bucket
+ all-reduce SUM).  The HIP model cannot run on CPU, so a small torch module with the SAME loss
convention (batch-summed relative L2, one parameter that receives no gradient) stands in: the
property under test — N-rank result == 1-rank result at the global batch — is model-independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(7)
        self.placeholder = torch.nn.Parameter(torch.rand(8))        # never used -> grad stays None
        self.a = torch.nn.Linear(6, 8)
        self.b = torch.nn.Linear(8, 1)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _loss(model, x, y):
    from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss
    bsz = x.shape[0]
    return TestLoss(size_average=False)(model(x).reshape(bsz, -1), y.reshape(bsz, -1))


def _data():
    g = torch.Generator().manual_seed(3)
    return torch.randn(8, 5, 6, generator=g), torch.randn(8, 5, 1, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformerbasednavierstokesolver_amd import ddp
    model = Toy()
    ddp.broadcast_parameters(model)
    x, y = ddp.shard_batch(_data(), rank, world)
    sync = ddp.FlatGradSync(model.parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=1e-2)
    for _ in range(2):                       # second step exercises the persistent flat-view path
        opt.zero_grad(set_to_none=False)
        _loss(model, x, y).backward()
        sync()
        opt.step()
    if rank == 0:
        torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, out)
        assert model.placeholder.grad is None and sync.nbytes == 4 * (6 * 8 + 8 + 8 + 1)
    dist.destroy_process_group()


def test_two_rank_training_equals_single_process(tmp_path):
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    ref = Toy()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=1e-2)
    x, y = _data()
    for _ in range(2):
        opt.zero_grad()
        _loss(ref, x, y).backward()
        opt.step()
    for k, v in ref.state_dict().items():
        assert torch.allclose(got[k], v, rtol=1e-5, atol=1e-7), k
    # the unused parameter saw neither gradient nor weight decay on either path
    assert torch.equal(got["placeholder"], Toy().placeholder.detach())


def test_shard_batch_rejects_uneven_split():
    from transformerbasednavierstokesolver_amd import ddp
    with pytest.raises(ValueError):
        ddp.shard_batch([torch.zeros(5, 2)], 0, 2)
