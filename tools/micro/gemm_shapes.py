"""The AT step's GEMM shapes on egz_gemm (exact-f32 MFMA): us per call.  EGZ_GEMM_TILE=0|1|2 forces the 64x64 / 64x32 / 32x32 tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import egaze_amd  # noqa
import egaze_amd.hipops as H


def timeit(fn, iters=30):
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
g = torch.Generator().manual_seed(1)
x = torch.randn(512, 512, generator=g).to(dev)
w4 = torch.randn(2048, 512, generator=g).to(dev)
w1 = torch.randn(512, 512, generator=g).to(dev)
dg = torch.randn(512, 2048, generator=g).to(dev)
b4 = torch.randn(2048, generator=g).to(dev)
res = [
    ("linear 512x512 . (2048x512)^T + b  (gx0)", timeit(lambda: H.linear_fwd(x, w4, bias=b4)), H.linear_fwd(x, w4, bias=b4), x @ w4.t() + b4),
    ("linear 512x512 . (512x512)^T + relu (lin)", timeit(lambda: H.linear_fwd(x, w1, relu=True)), H.linear_fwd(x, w1, relu=True), torch.relu(x @ w1.t())),
    ("matmul_tn (512x2048)^T . 512x512   (dW)  ", timeit(lambda: H.matmul_tn(dg, x)), H.matmul_tn(dg, x), dg.t() @ x),
    ("matmul_tn (512x512)^T . 512x512  (d lin) ", timeit(lambda: H.matmul_tn(x, x)), H.matmul_tn(x, x), x.t() @ x),
    ("matmul_nn 512x512 . 512x512     (dh_top) ", timeit(lambda: H.matmul_nn(x, w1)), H.matmul_nn(x, w1), x @ w1),
    ("matmul_nn 512x2048 . 2048x512   (dx)     ", timeit(lambda: H.matmul_nn(dg, w4)), H.matmul_nn(dg, w4), dg @ w4),
]
for name, us, got, ref in res:
    err = ((got.double() - ref.double()).abs().max() / ref.double().abs().max()).item()
    print(f"{name}: {us:6.1f} us   rel err vs torch {err:.1e}")
