"""world_size-2 gloo test of the data-parallel gradient exchange (dp.GradReducer) on CPU: bucketed in-place
all-reduce of a flat gradient buffer driven by post-accumulate-grad hooks == full-batch gradient."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import egaze_amd  # noqa: F401
    from egaze_amd.dp import GradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 50), torch.nn.ReLU(), torch.nn.Linear(50, 21),
                              torch.nn.ReLU(), torch.nn.Linear(21, 3))
    if rank == 1:                      # replicas start different: the broadcast must fix that
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    params = list(net.parameters())
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat_p, flat_g = torch.zeros(off), torch.zeros(off)
    with torch.no_grad():
        for p, o in zip(params, offsets):
            flat_p[o:o + p.numel()].copy_(p.reshape(-1))
            p.data = flat_p[o:o + p.numel()].view(p.shape)
            p.grad = flat_g[o:o + p.numel()].view(p.shape)
    red = GradReducer(flat_g, params, offsets, bucket_bytes=4096, flat_param=flat_p)
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 37, generator=g), torch.randn(8, 3, generator=g)
    out = []
    for step in range(2):              # two steps: counters must reset
        flat_g.zero_()
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        ((net(xs) - ys) ** 2).sum().backward()
        red.wait()
        out.append(flat_g.clone() * red.grad_scale)
    q.put((rank, flat_p.clone(), out))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1])                       # broadcast made replicas identical
    # reference: full-batch gradient / world on rank-0's parameters
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 50), torch.nn.ReLU(), torch.nn.Linear(50, 21),
                              torch.nn.ReLU(), torch.nn.Linear(21, 3))
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 37, generator=g), torch.randn(8, 3, generator=g)
    ((net(x) - y) ** 2).sum().backward()
    ref = []
    for p in net.parameters():
        ref.append(p.grad.reshape(-1) / world)
        pad = (-p.numel()) % 4
        if pad:
            ref.append(torch.zeros(pad))
    ref = torch.cat(ref)
    for rank in range(world):
        for step in range(2):
            assert torch.allclose(res[rank][2][step], ref, rtol=1e-5, atol=1e-6)


def test_rank_shard_sampler_partitions_and_pads():
    """Every rank sees a disjoint share, the same number of minibatches (the all-reduce is a collective), a new
    order per epoch; pad=False (validation) never repeats a sample."""
    sys.path.insert(0, ROOT)
    import egaze_amd  # noqa: F401
    from egaze_amd.dp import RankShardSampler
    data = list(range(103))
    for world, bs in ((2, 8), (8, 4), (3, 5)):
        sam = [RankShardSampler(data, True, bs, seed=7, world=world, rank_=r) for r in range(world)]
        shares = [list(s) for s in sam]
        assert len({len(s) for s in shares}) == 1 and len(shares[0]) % bs == 0 and len(sam[0]) == len(shares[0])
        flat = [i for s in shares for i in s]
        assert set(flat) == set(data) and len(flat) - len(data) < world * bs          # only the wrap-around padding repeats
        for s in sam:
            s.set_epoch(1)
        assert [list(s) for s in sam] != shares
        val = [list(RankShardSampler(data, False, bs, pad=False, world=world, rank_=r)) for r in range(world)]
        assert sorted(i for s in val for i in s) == data
    one = RankShardSampler(data, False, 8, world=1, rank_=0)
    assert list(one) == data
