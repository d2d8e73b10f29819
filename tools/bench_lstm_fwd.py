"""Forward recurrence of the AT network (nn.LSTM(512, 512, 2), T = 16, B = 32) alone: the T + 1 wavefront launches (egz_lstm_wave_fwd)
against the one persistent weight-stationary launch (egz_lstm_persist_fwd), us per sequence, alternating, each captured into a
hipGraph of 20 sequences (so neither is host-bound).  Usage: python tools/bench_lstm_fwd.py [--T 16] [--B 32]"""
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
bsum = [None, (torch.randn(2048, generator=g) * 0.1).to(dev)]
gx0 = torch.randn(a.T, a.B, 2048, generator=g).to(dev)
h0 = (torch.randn(2, a.B, 512, generator=g) * 0.5).to(dev)
c0 = (torch.randn(2, a.B, 512, generator=g) * 0.5).to(dev)
REP = 20


def build(persist):
    H.LSTM_PERSIST = persist
    for _ in range(3):
        H.lstm_wave_fwd(gx0, w_ih, w_hh, bsum, h0, c0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    keep = []
    with H.capture(graph):
        for _ in range(REP):
            keep.append(H.lstm_wave_fwd(gx0, w_ih, w_hh, bsum, h0, c0))
    return graph, keep


graphs = {p: build(p) for p in (False, True)}
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
        name = "persistent (1 launch)" if p else f"wavefront ({a.T + 1} launches)"
        print(f"round {rnd}  T={a.T} B={a.B}  {name:28s} {us:7.1f} us per sequence   {us / (a.T + 1):5.2f} us per global step", flush=True)
print("persist status word:", H.lstm_persist_status())
