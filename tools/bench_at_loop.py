"""AT.trainLSTM's per-fixation-sample loop (B = 1, T = 1; reference AT.py:127-145) on synthetic 512-vectors in host memory:
MSE(pred_{i-1}, tanh(target_i)) -> zero_grad -> backward -> Adam -> loss.item() -> repackage hidden -> forward.
Usage: python tools/bench_at_loop.py [--n 400]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import egaze_amd  # noqa: F401,E402
from egaze_amd.functions import MSELoss  # noqa: E402
from egaze_amd.models.LSTMnet import lstmnet  # noqa: E402
from egaze_amd.optim import FusedAdam  # noqa: E402
from egaze_amd.utils import repackage_hidden  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=400)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    lstm = lstmnet().to(dev)
    lstm.train()
    opt = FusedAdam(lstm.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(5)
    samples = [(torch.randn(1, 512, generator=g), torch.rand(1, 512, generator=g)) for _ in range(64)]

    def run(n):
        hidden, pred, last = None, None, 0.0
        stage = torch.empty((2, 2, 512)).pin_memory()          # as AT._epoch: one async copy per sample from a pinned ring
        for i in range(n):
            inp_h, gt_h = samples[i % 64]
            slot = stage[i & 1]
            slot[0].copy_(inp_h.reshape(-1))
            slot[1].copy_(gt_h.reshape(-1))
            both = slot.to(dev, non_blocking=True)
            inp, target = both[0].view(1, 1, -1), both[1].view(1, 1, -1)
            if pred is not None:
                loss = MSELoss.apply(pred, torch.tanh(target))
                opt.zero_grad()
                loss.backward()
                opt.step()
                last = loss.item()
            hidden = repackage_hidden(hidden)
            pred, hidden = lstm(inp, hidden)
        return last

    run(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run(args.n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.n
    print(f"AT.trainLSTM loop (B=1, T=1): {dt * 1e6:.0f} us per sample, {1 / dt:.0f} samples/s, last loss {last:.6f}")


if __name__ == "__main__":
    main()
