"""N>1 path on CPU: two gloo ranks, gradient arena all-reduce (the RCCL path on MI355X), run through
the emulated C ABI.  Equivalence: mean of per-rank gradients == gradient of the mean loss over the
concatenated batch, so after one fused clip+Adam step both ranks hold the single-process weights."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(5, 6, generator=g), torch.randn(5, 2, generator=g)


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "kan-tts_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from util import emulation

    from kantts.train.optim import ArenaAdam, ParamArena

    with emulation():
        net = _net()
        if rank == 1:  # replicas start different on purpose: enable_data_parallel must broadcast rank 0
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        arena = ParamArena(net)
        arena.enable_data_parallel(n_buckets=3)
        opt = ArenaAdam(arena, lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
        opt.set_grad_clip(0.5)
        for step in range(3):
            x, y = _data(rank + 10 * step)
            opt.zero_grad()
            ((net(x) - y) ** 2).mean().backward()
            opt.step()
        torch.save(arena.flat.clone(), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_arena_allreduce_equals_single_process(tmp_path):
    from util import emulation

    from kantts.train.optim import ArenaAdam, ParamArena

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(r0, r1)
    with emulation():
        net = _net()
        arena = ParamArena(net)
        opt = ArenaAdam(arena, lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
        opt.set_grad_clip(0.5)
        for step in range(3):
            xs, ys = zip(*[_data(r + 10 * step) for r in range(2)])
            x, y = torch.cat(xs), torch.cat(ys)
            opt.zero_grad()
            ((net(x) - y) ** 2).mean().backward()
            opt.step()
    assert float((arena.flat - r0).abs().max()) < 1e-6
