"""world_size-2 gloo test of the data-parallel exchange (CPU): shards of the batch on two ranks, gradients
summed over ranks and averaged inside the optimizer step, must reproduce the single-process step on every
rank (SURVEY 8e oracle: per-shard backward, average, one AdamW).  The compute on each rank is the CPU oracle;
what is under test is the host logic of the exchange (flat arena all-reduce + 1/world scaling, dead-parameter
exclusion, identical post-step weights on all ranks)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(3)
    import faoctasr
    # a small two-layer stand-in with a dead parameter (never receives a gradient), wrapped in the real ParamArena
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({"a": torch.nn.Conv2d(1, 4, 3, padding=1), "b": torch.nn.Conv2d(4, 1, 3, padding=1),
                               "dead": torch.nn.Conv2d(1, 1, 1)})
    live = [(n, p) for n, p in net.named_parameters() if not n.startswith("dead.")]
    arena = faoctasr.ParamArena(live, lr=1e-2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 8, 8, generator=g)
    y = torch.randn(4, 1, 8, 8, generator=g)
    shard = slice(rank * 2, rank * 2 + 2)
    arena.zero_grad()
    loss = torch.nn.functional.l1_loss(net["b"](torch.relu(net["a"](x[shard]))), y[shard])      # mean over the LOCAL shard
    loss.backward()
    arena.all_reduce()
    # AdamW with grad_scale = 1/world (what faoctasr_adamw_step does), restated on the host for the CPU test
    gavg = arena.grad / world
    ret[rank] = (gavg.clone(), float(loss), net["dead"].weight.grad is None)
    dist.destroy_process_group()


def test_two_rank_gradient_exchange():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2
    g0, g1 = ret[0][0], ret[1][0]
    assert torch.equal(g0, g1)                                   # every rank holds the same averaged gradient
    assert ret[0][2] and ret[1][2]                               # the dead parameter never entered the exchange
    # single-process reference on the full batch: equal shards => mean of shard means == global mean
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({"a": torch.nn.Conv2d(1, 4, 3, padding=1), "b": torch.nn.Conv2d(4, 1, 3, padding=1),
                               "dead": torch.nn.Conv2d(1, 1, 1)})
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 8, 8, generator=g)
    y = torch.randn(4, 1, 8, 8, generator=g)
    torch.nn.functional.l1_loss(net["b"](torch.relu(net["a"](x))), y).backward()
    ref = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for n, p in net.named_parameters() if not n.startswith("dead.")])
    assert torch.allclose(g0, ref, rtol=1e-5, atol=1e-7)


def _oracle_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(4)
    import random
    from oracle import octa_oracle as O
    random.seed(1234 + rank)
    S = O.StepOracle(seed=0)
    a, b = O.synthetic_batch(2, 192, seed=99)                      # global batch 2, one image per rank

    def hook(phase, params):                                       # the exchange the build inserts (train.py:238/239, 267/269)
        flat = torch.cat([p.grad.reshape(-1) for p in params if p.grad is not None])
        dist.all_reduce(flat)
        flat /= world
        o = 0
        for p in params:
            if p.grad is not None:
                p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
    L = S.train_step(a[rank:rank + 1], b[rank:rank + 1], grad_hook=hook)
    chk = {k: float(sum(p.detach().double().abs().sum() for p in ps)) for k, ps in S.params.items()}
    ret[rank] = (L["loss_G"], chk)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_full_step_replicas_stay_identical():
    """The whole train step on 2 ranks (one image each, CPU oracle compute, gloo): after the exchange + AdamW both ranks
    hold bit-identical weights, although their local losses differ (per-replica BatchNorm statistics: SURVEY 8e)."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_oracle_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][1] == ret[1][1]
    assert ret[0][0] != ret[1][0]
