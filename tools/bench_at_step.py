"""AT step alone at BASELINE config 4's shape (lstmnet T = 16, B = 32: forward + MSE + backward + Adam), ms per step: issued launch
by launch, and as one hipGraph replay.  EGAZE_LSTM_PERSIST=0: the recurrence as wavefront launches instead of the two persistent ones."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import egaze_amd  # noqa: E402,F401
from egaze_amd.functions import MSELoss  # noqa: E402
from egaze_amd.models.LSTMnet import lstmnet  # noqa: E402
from egaze_amd.optim import FusedAdam  # noqa: E402
from oracle import synth as synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--B", type=int, default=32)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lstm = lstmnet().to(dev)
    lstm.train()
    opt = FusedAdam(lstm.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(3)
    at_in = torch.randn(args.T, args.B, 512, generator=g).to(dev)
    at_tgt = torch.tanh(torch.randn(args.T, args.B, 512, generator=g)).to(dev)
    h0 = torch.zeros(2, args.B, 512, device=dev)
    c0 = torch.zeros(2, args.B, 512, device=dev)
    opt.zero_grad()

    def step():
        pred, _ = lstm(at_in, (h0, c0))
        loss = MSELoss.apply(pred, at_tgt)
        loss.backward()
        opt.step()
        opt.zero_grad()

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(f"AT step T={args.T} B={args.B} eager: {ms:.3f} ms  ({args.T * args.B / ms * 1e3:.0f} (t, b) samples/s)")

    # the same step as one hipGraph replay (graphs.GraphedTrainStep)
    from egaze_amd.graphs import GraphedTrainStep

    def forward_loss(x, tgt):
        pred, _ = lstm(x, (h0, c0))
        return MSELoss.apply(pred, tgt), pred
    gs = GraphedTrainStep(forward_loss, opt, (at_in, at_tgt))
    for _ in range(6):
        gs(at_in, at_tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gs(at_in, at_tgt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(f"AT step T={args.T} B={args.B} hipGraph replay: {ms:.3f} ms  ({args.T * args.B / ms * 1e3:.0f} (t, b) samples/s)")
    gs.close()


if __name__ == "__main__":
    main()
