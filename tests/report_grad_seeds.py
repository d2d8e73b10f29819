"""Report tool (not a test): which seeded 32 x 32 inputs are flip-free for the whole-model gradient checks of
tests/test_hip_model_sp.py, per summation order (EGAZE split-K on / off).  Run on the GPU box:
    python tests/report_grad_seeds.py 0 12"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import egaze_amd.hipops as H  # noqa: E402
import test_hip_model_sp as T  # noqa: E402


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    for splitk in (False, True):
        H.SPLITK = splitk
        for seed in range(lo, hi):
            try:
                a = T._full_grads_small(seed)
            except AssertionError as e:
                a = ("ASSERT", str(e)[:80])
            try:
                b = T._grads_vs_fp64(seed)
            except AssertionError as e:
                b = ("ASSERT", str(e)[:80])
            print(f"splitk={int(splitk)} seed={seed} full_grads_tight={a[0]} vs_fp64_tight={b[0]} | {a[1]} | {b[1]}", flush=True)


if __name__ == "__main__":
    main()
