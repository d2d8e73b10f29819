"""What the epilogue of the streamed forward kernel costs per layer shape: bias / bias + ReLU (+ abs-max) / bias + BatchNorm sums.
python tools/micro/epi_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import egaze_amd  # noqa
import egaze_amd.hipops as H


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = "cuda:0"
for C, K, S in [(64, 64, 224), (128, 128, 112), (256, 256, 56), (512, 512, 28)]:
    B = 32
    x = torch.randn(B, S, S, C, device=dev).relu_()
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    b = torch.randn(K, device=dev)
    H.absmax_of(x)
    wp, st = H.conv_weight(w, "fwd", H.F16X3, x, K)
    assert st
    res = []
    for epi in (H.EPI_BIAS, H.EPI_BIAS_RELU, H.EPI_BIAS_STATS):
        res.append(timeit(lambda: H.conv3x3_fwd(x, wp, b, K, epi=epi, dtype=H.F16X3, streamed=True)))
    print(f"{C:4d} -> {K:4d} @ {S:3d}: bias {res[0]:7.1f} us   bias + ReLU + abs-max {res[1]:7.1f} us   bias + BN sums {res[2]:7.1f} us")
