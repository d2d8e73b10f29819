"""How long does ONE replay of AT's captured per-sample step take on the device, without the host side of the loop (loader, pinned
staging copy, H2D copy, replay call)?  python tools/micro/at_replay_only.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import egaze_amd  # noqa
import egaze_amd.AT as at_mod
from egaze_amd.functions import MSELoss
from egaze_amd.models.LSTMnet import lstmnet
from egaze_amd.optim import FusedAdam

dev = torch.device("cuda:0")
torch.manual_seed(3)
lstm = lstmnet().to(dev)
lstm.train()
opt = FusedAdam(lstm.parameters(), lr=1e-4)
ring = torch.zeros(32, device=dev)
r = at_mod._GraphedSampleStep(lstm, MSELoss.apply, opt, dev, 512, ring=ring)
pair = torch.rand(2, 512).pin_memory()
for _ in range(8):
    r.step(pair)
torch.cuda.synchronize()
assert r.graph is not None
N = 3000
t0 = time.perf_counter()
for _ in range(N):
    r.graph.replay()
torch.cuda.synchronize()
print(f"replay only: {(time.perf_counter() - t0) / N * 1e6:.1f} us per sample")
t0 = time.perf_counter()
for _ in range(N):
    r.both.copy_(pair.view_as(r.both), non_blocking=True)
    r.graph.replay()
torch.cuda.synchronize()
print(f"H2D copy + replay: {(time.perf_counter() - t0) / N * 1e6:.1f} us per sample")
t0 = time.perf_counter()
for _ in range(N):
    r.graph.replay()
    if _ % 32 == 31:
        ring.cpu()
torch.cuda.synchronize()
print(f"replay + one read-back per 32: {(time.perf_counter() - t0) / N * 1e6:.1f} us per sample")
r.close()
