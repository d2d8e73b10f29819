"""Micro-benchmark of the MFMA conv kernels on the SP layer shapes (B=32): forward / dgrad / wgrad, HIP-event timed.
Usage: python tools/bench_conv.py [--iters N] [--only substring] [--batch B]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egaze_amd  # noqa
import egaze_amd.hipops as H

PEAK = 157.3
# (name, Cin, Cout, H(out), ups)
SHAPES = [
    ("enc3   64->64  @224", 64, 64, 224, False),
    ("enc7   64->128 @112", 64, 128, 112, False),
    ("enc10 128->128 @112", 128, 128, 112, False),
    ("enc14 128->256 @56 ", 128, 256, 56, False),
    ("enc17 256->256 @56 ", 256, 256, 56, False),
    ("enc24 256->512 @28 ", 256, 512, 28, False),
    ("enc27 512->512 @28 ", 512, 512, 28, False),
    ("enc34 512->512 @14 ", 512, 512, 14, False),
    ("dec12 512->256 @56u", 512, 256, 56, True),
    ("dec19 256->128 @112u", 256, 128, 112, True),
    ("dec24 128->64  @224u", 128, 64, 224, True),
    ("dec26  64->64  @224", 64, 64, 224, False),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tile", type=lambda v: int(v, 0), default=0)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--shape", action="append", default=[], help="extra shape C,K,H,B (plain conv)")
    ap.add_argument("--dtype", type=int, default=0, help="1 = f16 x3 forward / 2 = bf16 x3 (split-half kernels)")
    ap.add_argument("--zero", action="store_true", help="all-zero operands (DVFS probe: same instruction stream, no data toggling)")
    ap.add_argument("--relu", action="store_true", help="post-ReLU-like operands: x = relu(randn), dy masked the same way (half zeros)")
    ap.add_argument("--presplit", action="store_true", help="also time the pre-split forms: forward with the per-channel max / min "
                    "epilogue (+bound), forward / weight gradient over a pre-split x operand (pre)")
    ap.add_argument("--wflag", type=lambda v: int(v, 0), default=0, help="0x800 = per-tap wgrad kernel")
    a = ap.parse_args()
    dev = "cuda:0"
    B = a.batch
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    shapes = [(n, c, k, h, u, B) for n, c, k, h, u in SHAPES]
    if a.shape:
        shapes = []
        for sh in a.shape:
            c, k, h, b = (int(v) for v in sh.split(","))
            shapes.append((f"custom {c}->{k} @{h} B{b}", c, k, h, False, b))
    for name, C, K, Hh, ups, B in shapes:
        if a.only and a.only not in name:
            continue
        hin = Hh // 2 if ups else Hh
        x = torch.randn(B, hin, hin, C, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        bias = torch.randn(K, device=dev)
        dy = torch.randn(B, Hh, Hh, K, device=dev)
        if a.zero:
            x.zero_(); w.zero_(); dy.zero_()
        if a.relu:
            x.clamp_(min=0); dy.mul_((torch.randn_like(dy) > 0).float())
        flops = 2.0 * B * Hh * Hh * K * 9 * C
        fdt = a.dtype if (a.dtype and K % 64 == 0) else 0
        ddt = a.dtype if (a.dtype and C % 64 == 0) else 0
        # plain convs: the packing / kernel hipops picks for this geometry (streamed-weight kernel unless EGAZE_STREAMED=0)
        wp, fst = H.conv_weight(w, "ups_fwd" if ups else "fwd", fdt, x, K)
        wd, dst = H.conv_weight(w, "dgrad", ddt, dy, C)
        res = []
        if "fwd" in a.what:
            t = timeit(lambda: H.conv3x3_fwd(x, wp, bias, K, ups="phase" if ups else False,
                                             epi=H.EPI_BIAS_RELU if ups else H.EPI_BIAS_STATS,
                                             tile_flag=a.tile, dtype=fdt, streamed=fst), a.iters)
            res.append(("fwd", t))
        if "dgrad" in a.what:
            # the data gradient as the step issues it: hipops.conv3x3_dgrad (two MFMA products per MAC unless EGAZE_BWD_PRODUCTS=3),
            # for an upsample-fused conv the polyphase form straight to the low-res gradient (hipops.conv3x3_ups_dgrad)
            if ups and ddt:
                wu, ust = H.conv_weight(w, "ups_dgrad", ddt, dy, C)
                t = timeit(lambda: H.conv3x3_ups_dgrad(dy, wu, C, dtype=ddt, streamed=ust), a.iters)
            elif ddt:
                t = timeit(lambda: H.conv3x3_dgrad(dy, wd, C, dtype=ddt, streamed=dst), a.iters)
            else:
                t = timeit(lambda: H.conv3x3_fwd(dy, wd, None, C, ups=False, epi=H.EPI_BIAS, tile_flag=a.tile, dtype=ddt,
                                                 streamed=dst), a.iters)
            res.append(("dgrad", t))
        if "wgrad" in a.what:
            t = timeit(lambda: H.conv3x3_wgrad(x, dy, ups=ups, variant_flag=a.wflag), a.iters)
            res.append(("wgrad", t))
        if a.presplit and not ups and fdt == 1 and C % 64 == 0 and K % 64 == 0 and H.presplit_ok(B, Hh, Hh, C, K):
            # a valid pre-split image of relu(x) (timing only: RNE halves instead of the kernels' RTZ hi half)
            xr = x.clamp(min=0)
            am = H.absmax_of(xr)
            sc = 2.0 ** (13 - torch.frexp(H.absmax_value(am))[1].item())
            xs = xr * sc
            hi = xs.half()
            lo = (xs - hi.float()).half()
            pair = torch.cat((hi.view(B, hin, hin, C // 4, 4), lo.view(B, hin, hin, C // 4, 4)), dim=-1).contiguous()
            xp = pair.view(torch.float32).view(B, hin, hin, C)
            xp._egz_absmax, xp._egz_presplit = am, True
            xr._egz_absmax = am
            st2 = H.EPI_BIAS_STATS
            for tag, fn in (("fwd(relu x)", lambda: H.conv3x3_fwd(xr, wp, bias, K, epi=st2, dtype=1, streamed=fst)),
                            ("fwd+bound", lambda: H.conv3x3_fwd(xr, wp, bias, K, epi=st2, dtype=1, streamed=fst, want_bound=True)),
                            ("fwd pre", lambda: H.conv3x3_fwd(xp, wp, bias, K, epi=st2, dtype=1, streamed=fst, pre_in=True)),
                            ("fwd pre+bound", lambda: H.conv3x3_fwd(xp, wp, bias, K, epi=st2, dtype=1, streamed=fst, pre_in=True, want_bound=True)),
                            ("wgrad(relu x)", lambda: H.conv3x3_wgrad(xr, dy, precision="split_f16")),
                            ("wgrad pre", lambda: H.conv3x3_wgrad(xp, dy, precision="split_f16", x_pre=True))):
                res.append((tag, timeit(fn, a.iters)))
        line = f"{name}  {flops/1e9:7.1f} GF"
        for k, t in res:
            tf = flops / (t * 1e-3) / 1e12
            tot.setdefault(k, [0.0, 0.0])
            tot[k][0] += flops
            tot[k][1] += t
            line += f" | {k} {t*1e3:7.0f} us {tf:6.1f} TF {100*tf/PEAK:4.1f}%"
        print(line, flush=True)
    for k, (f, t) in tot.items():
        if t > 0:
            print(f"TOTAL {k}: {t:.2f} ms, {f/(t*1e-3)/1e12:.1f} TF ({100*f/(t*1e-3)/1e12/PEAK:.1f}% of f32 MFMA peak)")


if __name__ == "__main__":
    main()
