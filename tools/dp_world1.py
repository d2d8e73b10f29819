"""The data-parallel code path at world size 1 on one GPU: SP (+ AT) step with an RCCL group of ONE rank, legs
plain / group only / reducer on, alternating, N rounds.  Usage: python tools/dp_world1.py [--rounds 3] [--steps 8] [--no-at]
[--trace]: only the reducer-on leg (for rocprofv3 --hip-trace --kernel-trace; tools/dp_trace_summary.py reads the csv)."""
import argparse
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--no-at", action="store_true")
    ap.add_argument("--trace", action="store_true")
    a = ap.parse_args()
    import torch.distributed as dist
    import egaze_amd  # noqa: F401
    from egaze_amd import dp, streams, synthetic
    from egaze_amd.floss import floss
    from egaze_amd.functions import MSELoss
    from egaze_amd.models.LSTMnet import lstmnet
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.optim import FusedAdam
    from egaze_amd.utils import cfg, make_layers
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(1234)
    model = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev).train()
    crit = floss().to(dev)
    opt = FusedAdam(model.parameters(), lr=1e-7)
    b = synthetic.sp_batch(32, 224, dev, seed=100)
    use_at = not a.no_at
    if use_at:
        lstm = lstmnet().to(dev).train()
        opt_at = FusedAdam(lstm.parameters(), lr=1e-4)
        atb = synthetic.at_batch(16, 32, dev, seed=200)
        at_in, at_tgt = atb["input"], torch.tanh(atb["gt"])
        h0 = torch.zeros(2, 32, 512, device=dev)
        c0 = torch.zeros(2, 32, 512, device=dev)
        opt_at.zero_grad()
        at_stream = streams.side_stream("at")
        at_stream.wait_stream(torch.cuda.current_stream())

    def step():
        out = model(b["image"], b["flow"])
        loss = crit(out, b["gt"].view(out.size()))
        loss.backward()
        opt.step()
        opt.zero_grad()
        if use_at:
            with torch.cuda.stream(at_stream):
                pred, _ = lstm(at_in, (h0, c0))
                MSELoss.apply(pred, at_tgt).backward()
                opt_at.step()
                opt_at.zero_grad()

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    opt.zero_grad()
    for _ in range(3):
        step()
    res = {"plain": [], "group": [], "reducer": []}
    if not a.trace:
        res["plain"].append(timed(a.steps))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    dist.barrier()
    for r in range(a.rounds):
        if not a.trace:
            step()
            res["group"].append(timed(a.steps))
        reds = [(dp.attach(opt, force=True), opt)] + ([(dp.attach(opt_at, force=True), opt_at)] if use_at else [])
        step(); step()
        res["reducer"].append(timed(a.steps))
        st = reds[0][0].stats
        for red, o in reds:
            red.detach(o)
        if a.trace:
            break
    dist.destroy_process_group()
    fmt = lambda v: " ".join(f"{x:.3f}" for x in v)
    print(f"plain {fmt(res['plain'])} | group only {fmt(res['group'])} | reducer on "
          f"{fmt(res['reducer'])} ms per step; buckets in backward {st['launched_in_backward']} / steps {st['steps']}")


if __name__ == "__main__":
    main()
