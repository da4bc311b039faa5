"""CPU: the data-parallel wiring (parallel.py, train_utils.py) over gloo with world_size 2.

The device ops need a GPU, so the ranks train a small stand-in module with the same forward(batch, return_loss)
contract; what is checked is the distributed logic the hot path relies on: one flat all-reduce per step, gradients
equal to the single-process emulation (shards run one after the other, gradients averaged), replicas identical
after the optimizer step, sharding helpers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import helpers

helpers.load_pkg()
from pn2_amd import parallel, train_utils  # noqa: E402


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = nn.Sequential(nn.Linear(5, 16), nn.Tanh(), nn.Linear(16, 3))

    def forward(self, batch, return_loss):
        out = self.net(batch["x"])
        loss = (out - batch["y"]).pow(2).mean()
        return (loss, {"offset_loss": loss.detach(), "semantic_loss": torch.zeros(())}) if return_loss else out


def make_data(n=64):
    g = torch.Generator().manual_seed(1)
    return torch.randn(n, 5, generator=g), torch.randn(n, 3, generator=g)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    model = Toy()
    sync = parallel.FlatGradAllReduce(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    x, y = make_data()
    lo, hi = parallel.shard_range(len(x), rank, world)
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    dist.all_reduce = counting
    sync.zero()
    # two backward passes (like the streaming mode's per-mini-batch backwards), ONE collective
    mid = (lo + hi) // 2
    for a, b in ((lo, mid), (mid, hi)):
        loss, _ = model({"x": x[a:b], "y": y[a:b]}, True)
        (loss * (b - a) / (hi - lo)).backward()
    sync.allreduce()
    assert calls["n"] == 1
    grad = sync.packed()
    opt.step()
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    if rank == 0:
        out["grad"] = grad.numpy()
        out["same"] = bool(all(torch.equal(gathered[0], g) for g in gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_matches_single_process_emulation():
    world, port = 2, free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
        grad, same = out["grad"], out["same"]
    assert same, "replicas diverged after the optimizer step"
    # emulation: run the shards sequentially in one process and average the gradients
    model = Toy()
    x, y = make_data()
    acc = None
    for r in range(world):
        lo, hi = parallel.shard_range(len(x), r, world)
        model.zero_grad()
        loss, _ = model({"x": x[lo:hi], "y": y[lo:hi]}, True)
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        acc = g if acc is None else acc + g
    np.testing.assert_allclose(grad, (acc / world).numpy(), rtol=1e-5, atol=1e-7)


def test_flat_views_survive_zero_and_reject_set_to_none():
    model = Toy()
    sync = parallel.FlatGradAllReduce(model)
    loss, _ = model({"x": torch.ones(4, 5), "y": torch.zeros(4, 3)}, True)
    loss.backward()
    assert float(sync.flat.abs().sum()) > 0
    sync.zero()
    assert float(sync.flat.abs().sum()) == 0 and all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in model.parameters())
    torch.optim.SGD(model.parameters(), lr=0.1).zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        sync.zero()


def test_sharding_helpers():
    spans = [parallel.shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
    costs = [6112, 1, 249, 658, 3000, 2999, 10, 10]          # raster sizes of one tree are this uneven
    shards = parallel.balanced_shards(costs, 2)
    assert sorted(i for s in shards for i in s) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= 0.1 * sum(costs)


def test_train_loop_single_process_cpu():
    """train()/validate() keep the reference's call pattern (scheduler.step(epoch) per batch, 50x loss, clip at 1.0)."""
    model = Toy()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=5)
    x, y = make_data()
    loader = [{"x": x[i:i + 16], "y": y[i:i + 16]} for i in range(0, 64, 16)]
    scaler = torch.amp.GradScaler("cpu", enabled=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        first = train_utils.train(model, loader, opt, sched, scaler, 0, None, False, False)
        for e in range(1, 6):
            last = train_utils.train(model, loader, opt, sched, scaler, e, None, False, False)
        val = train_utils.validate(model, loader, 0, None, False, False)
    assert last[0] < first[0] and np.isfinite(val[0])
