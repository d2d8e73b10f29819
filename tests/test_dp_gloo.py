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
    red = GradReducer(flat_g, params, offsets, bucket_bytes=4096, flat_param=flat_p, geometric=False)
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


def test_bucket_sizes_halve_along_the_backward_pass():
    """dp.GradReducer's default bucket policy: walking the parameters in backward order a bucket closes when it holds
    max(bucket_bytes, half of the gradient bytes still to come) -- few hand-overs, a small last bucket.  Every parameter is in
    exactly one bucket, buckets are contiguous and in backward order; ``geometric=False`` gives equal buckets."""
    sys.path.insert(0, ROOT)
    import egaze_amd  # noqa: F401
    from egaze_amd.dp import GradReducer
    sizes = [1728, 64, 36864, 64] + [589824, 256] * 12 + [2359296, 512] * 10 + [64, 1]      # an SP-like size profile, ~31 M floats
    params = [torch.nn.Parameter(torch.empty(n)) for n in sizes]
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(off)
    for geometric in (True, False):
        red = GradReducer(flat, params, offsets, bucket_bytes=8 << 20, geometric=geometric)
        assert not red.active                                   # no process group: no hooks, but the buckets are cut
        b = red.buckets
        assert b[0][1] == off and b[-1][0] == 0 and all(b[i][0] == b[i + 1][1] for i in range(len(b) - 1))
        assert sum(x[2] for x in b) == len(params) and set(red.bucket_of) == set(range(len(params)))
        nbytes = [(e - s_) * 4 for s_, e, _ in b]
        if geometric:
            assert nbytes[0] >= off * 4 // 2 and len(b) <= 6
            assert all(nbytes[i] >= nbytes[i + 1] * 0.6 for i in range(len(b) - 2))     # (non-increasing up to parameter granularity)
        else:
            assert len(b) >= 12 and max(nbytes) < (8 << 20) + 4 * 2359296
        print("geometric" if geometric else "equal", [round(v / 2 ** 20, 1) for v in nbytes])


def test_real_sp_parameter_list_buckets():
    """the REAL model_SP parameter list (215 state-dict entries, 134 trainable tensors, 46.5 M floats), laid out
    as FusedAdam lays it out (16-byte aligned slots in registration order), cut by the default policy: four in-place all-reduce
    buckets in backward order (decoder first), whatever the world size -- the cut depends on the
    parameter list only, so every rank of an 8-GPU run issues the same four collectives in the same order.  (Exact sizes: 92.1 +
    47.2 + 27.0 + 11.2 MiB = 96.6 + 49.4 + 28.3 + 11.7 MB of the 186.1 MB of gradients; "93 + 47 + 25 + 21" in the round-5 notes was
    the halving rule before parameter granularity: the 9.4 MB fusion / 512-channel tensors do not split.)"""
    sys.path.insert(0, ROOT)
    import egaze_amd  # noqa: F401
    from egaze_amd.dp import GradReducer
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.utils import make_layers, cfg
    model = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20))
    params = [p for p in model.parameters() if p.requires_grad]
    assert sum(p.numel() for p in params) == 46529409 and len(params) == 134
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    red = GradReducer(torch.empty(0), params, offsets)                  # (bucket cutting reads sizes only; no process group)
    nbytes = [(e - s_) * 4 for s_, e, _ in red.buckets]
    assert nbytes == [96623376, 49449216, 28329984, 11715072] and sum(nbytes) == off * 4 == 186117648, nbytes
    assert [b[2] for b in red.buckets] == [42, 48, 12, 32]              # tensors per bucket
    assert red.buckets[0][1] == off and red.buckets[-1][0] == 0
    names = {id(p): n for n, p in model.named_parameters()}
    first = {names[id(params[i])].split(".")[0] for i, b in red.bucket_of.items() if b == 0}
    last = {names[id(params[i])].split(".")[0] for i, b in red.bucket_of.items() if b == len(red.buckets) - 1}
    # backward order: the decoder's gradients fill the first bucket (with bn / fusion), the encoders' first layers the last one
    assert "decoder" in first and last <= {"features_s", "features_t"}, (first, last)


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


def _sink_worker(rank, world, port, q):
    """Gradient sinks (hipops.GradSink: the backward writes p.grad in place and returns None to autograd) under the
    reducer: a bucket's all-reduce must start only after EVERY parameter of the bucket has its final gradient, although
    torch also runs the post-accumulate hook for the None gradients (each parameter then reports twice)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import egaze_amd  # noqa: F401
    import egaze_amd.hipops as H
    from egaze_amd.dp import GradReducer

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w, b)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, g):
            x, w, b = ctx.saved_tensors
            dw, db = g.t() @ x, g.sum(0)
            sw, sb = H.grad_sink(w), H.grad_sink(b)
            if sw is not None:
                sw.view(w.shape).copy_(dw)
                H.grad_done(w)
                dw = None
            if sb is not None:
                sb.copy_(db)
                H.grad_done(b)
                db = None
            return g @ w, dw, db

    torch.manual_seed(0)
    dims = [37, 50, 21, 16, 9, 3]
    ps = []
    for a, b_ in zip(dims[:-1], dims[1:]):
        ps += [torch.nn.Parameter(torch.randn(b_, a) * 0.3), torch.nn.Parameter(torch.randn(b_) * 0.1)]
    offsets, off = [], 0
    for p in ps:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat_p, flat_g = torch.zeros(off), torch.zeros(off)

    class Owner:
        zero_gen = 0
    owner = Owner()
    with torch.no_grad():
        for p, o in zip(ps, offsets):
            flat_p[o:o + p.numel()].copy_(p.reshape(-1))
            p.data = flat_p[o:o + p.numel()].view(p.shape)
            p.grad = flat_g[o:o + p.numel()].view(p.shape)
            p._egz_sink = H.GradSink(flat_g[o:o + p.numel()], owner)
    x = torch.randn(4, 37, generator=torch.Generator().manual_seed(10 + rank))

    def fwd_bwd():
        flat_g.zero_()
        owner.zero_gen += 1
        h = x
        for i in range(0, len(ps), 2):
            h = Lin.apply(h, ps[i], ps[i + 1]).relu()
        h.sum().backward()

    fwd_bwd()
    g_local = flat_g.clone()
    gathered = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(gathered, g_local)
    red = GradReducer(flat_g, ps, offsets, bucket_bytes=2048, flat_param=flat_p, geometric=False)
    early = []
    launch = red._launch

    def checked_launch(b):
        s_, e_, _ = red.buckets[b]
        early.append(not torch.equal(flat_g[s_:e_], g_local[s_:e_]))      # launched before the bucket was complete?
        launch(b)
    red._launch = checked_launch
    fwd_bwd()
    red.wait()
    q.put((rank, len(red.buckets), any(early), torch.equal(flat_g, gathered[0] + gathered[1])))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_sinks_report_once_per_bucket():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sink_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_buckets, early, summed in res:
        assert n_buckets >= 3
        assert not early, "a bucket was all-reduced before all of its gradients had landed"
        assert summed


def _wait_worker(rank, world, port, q, fail):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import egaze_amd  # noqa: F401
    from egaze_amd.gaze_full import _wait_for_rank0
    t0 = time.time()
    try:
        if rank == 0:
            time.sleep(1.0)                                    # the sequential rank-0 stage (AT training / extraction)
            if fail:
                dist.distributed_c10d._get_default_store().set("stage/failed", "1")
                q.put((rank, "failed-flag-set", time.time() - t0))
                return
        _wait_for_rank0("stage", poll_s=0.1)
        q.put((rank, "released", time.time() - t0))
    except RuntimeError as e:
        q.put((rank, "raised: " + str(e), time.time() - t0))


@pytest.mark.parametrize("fail", [False, True])
def test_wait_for_rank0_is_a_host_side_rendezvous(fail):
    """gaze_full's wait for the rank-0-only AT stage: a store key polled on the host (no device collective, no multi-day
    timeout).  Rank 1 is released once rank 0 sets the key -- and is ENDED, not left hanging, when rank 0 fails."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wait_worker, args=(r, 2, port, q, fail)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(2):
        r, what, dt = q.get(timeout=120)
        res[r] = (what, dt)
    for p in procs:
        p.join(30)
    if fail:
        assert res[1][0].startswith("raised: rank 0 failed"), res
    else:
        assert res[0][0] == "released" and res[1][0] == "released" and res[1][1] >= 0.9, res


def _forced_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import egaze_amd  # noqa: F401
    from egaze_amd.dp import GradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 50), torch.nn.ReLU(), torch.nn.Linear(50, 3))
    params = list(net.parameters())
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat_g = torch.zeros(off)
    for p, o in zip(params, offsets):
        p.grad = flat_g[o:o + p.numel()].view(p.shape)
    x = torch.randn(4, 37)
    (net(x) ** 2).sum().backward()
    plain = flat_g.clone()
    idle = GradReducer(flat_g, params, offsets, bucket_bytes=512)              # world 1, not forced: a no-op
    assert not idle.active
    red = GradReducer(flat_g, params, offsets, bucket_bytes=512, force=True, geometric=False)
    assert red.active and red.world == 1 and len(red.buckets) >= 2
    flat_g.zero_()
    (net(x) ** 2).sum().backward()
    in_bwd = red.stats["launched_in_backward"]
    red.wait()
    q.put((torch.equal(flat_g, plain), in_bwd, red.stats["launched_in_wait"], len(red.buckets), red.grad_scale))
    red.detach()
    assert not red.active


def test_reducer_forced_at_world_one():
    """dp.GradReducer(force=True) at world size 1 (how the RCCL path is exercised on a one-GPU box): hooks fire, every bucket
    is all-reduced (a 1-rank sum = identity), gradients unchanged, scale 1."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(0, 1, _free_port(), q))
    p.start()
    same, in_bwd, in_wait, nb, scale = q.get(timeout=120)
    p.join(30)
    assert same and in_bwd + in_wait == nb and in_bwd >= 1 and scale == 1.0
