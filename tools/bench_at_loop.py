"""AT.trainLSTM's per-fixation-sample loop (B = 1, T = 1; reference AT.py:127-145) on synthetic 512-vectors in host memory,
through the driver's own AT._epoch: hipGraph replay per sample (default), launch by launch (EGAZE_AT_GRAPH=0), and launch by
launch on the sequence kernels (EGAZE_AT_GRAPH=0 EGAZE_LSTM_B1=0).  Usage: python tools/bench_at_loop.py [--n 2000]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import egaze_amd  # noqa: F401,E402
import egaze_amd.AT as at_mod  # noqa: E402
from egaze_amd.functions import MSELoss  # noqa: E402
from egaze_amd.models.LSTMnet import lstmnet  # noqa: E402
from egaze_amd.optim import FusedAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    lstm = lstmnet().to(dev)
    lstm.train()
    at = at_mod.AT.__new__(at_mod.AT)            # the constructor wants an SP checkpoint and dataset folders
    at.lstm, at.criterion_lstm, at.device = lstm, MSELoss.apply, dev
    at.optimizer_lstm = FusedAdam(lstm.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(5)
    base = [{"input": torch.randn(1, 512, generator=g), "gt": torch.rand(1, 512, generator=g),
             "same": torch.tensor([0 if i % 37 == 36 else 1])} for i in range(64)]
    at._epoch([base[i % 64] for i in range(50)], True)
    torch.cuda.synchronize()
    loader = [base[i % 64] for i in range(args.n)]
    t0 = time.perf_counter()
    last = at._epoch(loader, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.n
    mode = "hipGraph replay" if at_mod.AT_GRAPH else "launch by launch"
    print(f"AT.trainLSTM loop (B=1, T=1, {mode}): {dt * 1e6:.0f} us per sample, {1 / dt:.0f} samples/s, mean loss {last:.6f}")


if __name__ == "__main__":
    main()
