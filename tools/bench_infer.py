"""SP inference latency at small batch (eval-mode model_SP forward, the reference's extraction / run_spatialstream schedule):
eager launches vs one captured hipGraph replay (egaze_amd.graphs.GraphedModule).  Usage: python tools/bench_infer.py [batches, e.g. 1,4,32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egaze_amd  # noqa
from egaze_amd import synthetic
from egaze_amd.graphs import GraphedModule
from egaze_amd.models.model_SP import model_SP
from egaze_amd.utils import cfg, make_layers

dev = torch.device("cuda", 0)
torch.manual_seed(0)
sp = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev).eval()


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n, (t2 - t0) / n


for B in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("1", "4", "32"))]:
    b = synthetic.sp_batch(B, 224, dev, seed=1)
    with torch.no_grad():
        ref = sp(b["image"], b["flow"])
        issue, total = timed(lambda: sp(b["image"], b["flow"]), 30)
        g = GraphedModule(sp, (b["image"], b["flow"]))
        out = g(b["image"], b["flow"])
        same = torch.equal(out, ref)
        gi, gt = timed(lambda: g(b["image"], b["flow"]), 30)
    print(f"SP inference B={B}: eager {total*1e3:.2f} ms (host issue {issue*1e3:.2f} ms), graph replay {gt*1e3:.2f} ms "
          f"(host {gi*1e3:.3f} ms), {B/gt:.0f} frames/s, identical output: {same}")
