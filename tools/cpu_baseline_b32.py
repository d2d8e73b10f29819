"""One-off measurement for profiles/: the oracle's SP train step (the reference's PyTorch-CPU algorithm) at the headline batch
32 and at the bench's bounded batch 8, same thread count -- replaces the unverified 'frames/s is batch-independent' sentence
(VERDICT r2 weak #6)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import egaze_oracle as O  # noqa: E402
from oracle import synth  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(threads)
    print(f"host threads used: {threads} of {os.cpu_count()}")
    for batch, steps in ((8, 3), (32, 2)):
        sd = synth.synth_state_dict(O.sp_shapes(), seed=1, head_gain=0.25)
        x_s, x_t, gt, _ = synth.synth_sp_batch(batch, 224, seed=0)
        opt = {}
        O.sp_train_step(sd, opt, 1, x_s, x_t, gt, 1e-7)
        t0 = time.perf_counter()
        for i in range(steps):
            O.sp_train_step(sd, opt, 2 + i, x_s, x_t, gt, 1e-7)
        dt = (time.perf_counter() - t0) / steps
        print(f"batch {batch}: {dt:.2f} s/step = {batch / dt:.2f} frames/s (1 warm-up + {steps} timed steps)", flush=True)


if __name__ == "__main__":
    main()
