"""The recurrence of the AT network (nn.LSTM(512, 512, 2), T = 16, B = 32) alone, forward and backward: the wavefront launches
(egz_lstm_wave_fwd: T + 1, egz_lstm_wave_bwd: T + 3) against the one persistent weight-stationary launch per direction
(egz_lstm_persist_fwd / _bwd), us per sequence, alternating, each captured into a hipGraph of 20 sequences (so neither is
host-bound).  Usage: python tools/bench_lstm_seq.py [--T 16] [--B 32]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import egaze_amd  # noqa: E402,F401
from egaze_amd import hipops as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=16)
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
w_ih = [None, (torch.randn(2048, 512, generator=g) * 0.05).to(dev)]
w_hh = [(torch.randn(2048, 512, generator=g) * 0.05).to(dev) for _ in range(2)]
b_ih = [(torch.randn(2048, generator=g) * 0.1).to(dev) for _ in range(2)]
b_hh = [(torch.randn(2048, generator=g) * 0.1).to(dev) for _ in range(2)]
bsum = [a_ + b_ for a_, b_ in zip(b_ih, b_hh)]
db = [torch.empty(2048, device=dev) for _ in range(4)]
w_hh_t = [w.t().contiguous() for w in w_hh]
w_ih_t = [None, w_ih[1].t().contiguous()]
gx0 = torch.randn(a.T, a.B, 2048, generator=g).to(dev)
h0 = (torch.randn(2, a.B, 512, generator=g) * 0.5).to(dev)
c0 = (torch.randn(2, a.B, 512, generator=g) * 0.5).to(dev)
dh_top = torch.randn(a.T, a.B, 512, generator=g).to(dev)
hs, cs, acts, hn, cn = H.lstm_wave_fwd(gx0, w_ih, w_hh, bsum, h0, c0)
REP = 20


def build(persist, direction):
    if direction == "fwd":
        call = (lambda: H.lstm_persist_fwd(gx0, w_ih, w_hh, b_ih, b_hh, h0, c0)) if persist else \
               (lambda: H.lstm_wave_fwd(gx0, w_ih, w_hh, bsum, h0, c0))
    else:          # the persistent launch also forms the four bias gradients (the wavefront path needs 2 x egz_colsum + 2 copies more)
        call = (lambda: H.lstm_persist_bwd(dh_top, None, None, acts, cs, c0, w_hh, w_ih, db)) if persist else \
               (lambda: H.lstm_wave_bwd(dh_top, None, None, acts, cs, c0, w_hh_t, w_ih_t))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    keep = []
    with H.capture(graph):
        for _ in range(REP):
            keep.append(call())
    return graph, keep


for direction, nl in (("fwd", a.T + 1), ("bwd", a.T + 3)):
    graphs = {p: build(p, direction) for p in (False, True)}
    for rnd in range(a.rounds):
        for p in (False, True):
            gr = graphs[p][0]
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                gr.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (10 * REP) * 1e6
            name = "persistent (1 launch)" if p else f"wavefront ({nl} launches)"
            steps = (a.T + 1 if direction == "fwd" else a.T + 2) if p else nl
            print(f"round {rnd}  {direction}  T={a.T} B={a.B}  {name:28s} {us:7.1f} us per sequence   {us / steps:5.2f} us per step", flush=True)
    del graphs
print("persist status word:", H.lstm_persist_status())
