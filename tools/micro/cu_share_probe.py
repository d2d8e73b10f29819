"""What does the SP step's conv kernel lose on fewer CUs?  A blocker kernel (tools/micro/cu_blocker.hip) holds N CUs (whole-LDS
blocks that sleep) on a side stream while the conv microbenchmark runs on the others.  Usage: python tools/micro/cu_share_probe.py"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import egaze_amd  # noqa
import egaze_amd.hipops as H

blk = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libcu_blocker.so"))
blk.cu_blocker_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
side = torch.cuda.Stream()
where = torch.zeros(2 * 256, dtype=torch.int32, device=dev)


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


SHAPES = [("enc10 128->128 @112", 128, 128, 112), ("enc27 512->512 @28", 512, 512, 28), ("dec26 64->64 @224", 64, 64, 224)]
B = 32
print("us per launch (forward, f16x3 streamed kernel, post-ReLU operand / weight gradient) with N CUs held by the blocker")
for name, C, K, Hh in SHAPES:
    x = torch.randn(B, Hh, Hh, C, device=dev).clamp_(min=0)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    bias = torch.randn(K, device=dev)
    dy = torch.randn(B, Hh, Hh, K, device=dev) * 1e-3
    wp, st = H.conv_weight(w, "fwd", 1, x, K)
    line = f"{name}:"
    for nblk in (0, 16, 32, 48, 64):
        torch.cuda.synchronize()
        if nblk:
            with torch.cuda.stream(side):
                rc = blk.cu_blocker_launch(nblk, int(0.15 * 1.0e8), where.data_ptr(), torch.cuda.current_stream().cuda_stream)   # 0.15 s of the 100 MHz clock
                assert rc == 0, rc
            time.sleep(0.02)
        tf = timeit(lambda: H.conv3x3_fwd(x, wp, bias, K, epi=H.EPI_BIAS_STATS, dtype=1, streamed=st))
        tw = timeit(lambda: H.conv3x3_wgrad(x, dy, precision="split_f16"))
        torch.cuda.synchronize()
        if nblk:
            hw = where[:2 * nblk].cpu().view(-1, 2)
            cus = {(int(a[1]) & 15, int(a[0]) & 0xff00) for a in hw}
            line += f" | {nblk} held ({len(cus)} distinct CUs): fwd {tf:.0f} wgrad {tw:.0f}"
        else:
            line += f" 0 held: fwd {tf:.0f} wgrad {tw:.0f}"
    print(line, flush=True)
