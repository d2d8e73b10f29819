"""GPU parity tests, per C-ABI op: HIP kernel vs the same op in plain torch-CPU fp32 (the oracle's
building blocks) on seeded inputs.  Tolerance: max|d|/max|ref| <= 2e-5 for fp32 contractions (only the
summation order differs from the reference), bit-exact where the arithmetic is order-free."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def H():
    import egaze_amd.hipops as h
    return h


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def nhwc(t):   # (B,C,H,W) cpu -> (B,H,W,C) cuda contiguous
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):   # (B,H,W,C) cuda -> (B,C,H,W) cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("B,Hh,Ww,C,K,tile", [
    (2, 12, 12, 64, 64, 0),          # ragged M (288 = 2*128 + 32), BN=64
    (1, 14, 14, 128, 128, 0x200),    # BN=128 forced
    (3, 9, 7, 32, 192, 0),           # odd sizes, K=192 -> BN=64
    (2, 16, 16, 256, 256, 0x200),
])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_conv3x3_fwd(B, Hh, Ww, C, K, tile, epi):
    h = H()
    x = rnd(B, C, Hh, Ww, seed=1)
    w = rnd(K, C, 3, 3, seed=2, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=3, scale=0.1)
    ref = F.conv2d(x, w, b, padding=1)
    if epi == 1:
        ref = F.relu(ref)
    wd = w.to(DEV)
    y, stat = h.conv3x3_fwd(nhwc(x), h.packed_weight(wd, "fwd"), b.to(DEV), K, ups=False, epi=epi, tile_flag=tile)
    assert rel(nchw(y), ref) < 2e-5
    if epi == 2:
        s = stat.sum(0).cpu()                       # (2, K) fp64
        assert rel(s[0], ref.double().sum(dim=(0, 2, 3))) < 1e-5
        assert rel(s[1], (ref.double() ** 2).sum(dim=(0, 2, 3))) < 1e-5


@pytest.mark.parametrize("mode", ["fold", "phase"])
@pytest.mark.parametrize("B,Hh,Ww,C,K", [(2, 6, 6, 64, 64), (1, 7, 5, 128, 128), (3, 9, 4, 64, 256)])
def test_conv3x3_upsample_fused(B, Hh, Ww, C, K, mode):
    """[nearest x2 upsample -> conv3x3 -> ReLU] without materialising the upsampled tensor: folded gather (9 taps on
    the virtual image) and the phase decomposition (four 2x2 convs with pre-summed weights, 4/9 of the MACs)."""
    h = H()
    x = rnd(B, C, Hh, Ww, seed=4)
    w = rnd(K, C, 3, 3, seed=5, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=6, scale=0.1)
    pre = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    wd = w.to(DEV)
    wp = h.packed_weight(wd, "fwd" if mode == "fold" else "ups_fwd")
    y, _ = h.conv3x3_fwd(nhwc(x), wp, b.to(DEV), K, ups=mode, epi=1)
    assert tuple(y.shape) == (B, 2 * Hh, 2 * Ww, K)
    assert rel(nchw(y), F.relu(pre)) < 2e-5
    y2, stat = h.conv3x3_fwd(nhwc(x), wp, b.to(DEV), K, ups=mode, epi=2)
    assert rel(nchw(y2), pre) < 2e-5
    assert rel(stat.sum(0)[0].cpu(), pre.double().sum(dim=(0, 2, 3))) < 1e-5
    assert rel(stat.sum(0)[1].cpu(), (pre.double() ** 2).sum(dim=(0, 2, 3))) < 1e-5


@pytest.mark.parametrize("B,Hh,Ww,C,K,ups", [
    (2, 12, 12, 64, 64, False), (1, 14, 14, 128, 256, False), (2, 10, 6, 192, 64, False),
    (2, 12, 12, 64, 128, True), (1, 8, 8, 128, 128, True),
    # geometries of the 9-tap fused wgrad kernel (row segments of 32 / 28 / 14 pixels)
    (2, 28, 28, 64, 64, False), (1, 32, 32, 64, 128, False), (1, 28, 28, 128, 64, True), (3, 6, 28, 64, 64, False),
    (1, 16, 64, 64, 64, True), (2, 14, 14, 192, 128, False), (1, 28, 56, 64, 64, False),
    (1, 56, 56, 64, 64, True), (2, 6, 56, 128, 64, True),        # phase-decomposed upsample wgrad, L = 28
])
def test_conv3x3_backward(B, Hh, Ww, C, K, ups):
    """dgrad (same kernel, tap-flipped transposed weights), wgrad (split-K MFMA), bias grad, upsample bwd."""
    h = H()
    hin, win = (Hh // 2, Ww // 2) if ups else (Hh, Ww)
    x = rnd(B, C, hin, win, seed=7).requires_grad_(True)
    w = rnd(K, C, 3, 3, seed=8, scale=(2.0 / (9 * C)) ** 0.5).requires_grad_(True)
    b = rnd(K, seed=9, scale=0.1).requires_grad_(True)
    dy = rnd(B, K, Hh, Ww, seed=10)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    F.conv2d(xin, w, b, padding=1).backward(dy)
    dyd = nhwc(dy)
    wdev = w.detach().to(DEV)
    dxu = h.conv3x3_dgrad(dyd, h.packed_weight(wdev, "dgrad"), C)
    dx = h.upsample2x_bwd(dxu) if ups else dxu
    assert rel(nchw(dx), x.grad) < 2e-5
    if ups:      # fused form: 4x4 / stride-2 gather straight to the low-res gradient
        assert rel(nchw(h.conv3x3_ups_dgrad(dyd, h.packed_weight(wdev, "ups_dgrad"), C)), x.grad) < 2e-5
    dw = h.conv3x3_wgrad(nhwc(x.detach()), dyd, ups=ups)
    assert rel(dw.cpu(), w.grad) < 2e-5
    assert rel(h.colsum(dyd).cpu(), b.grad) < 1e-5


@pytest.mark.parametrize("B,Hh,Ww,C,K,ups", [
    (2, 8, 32, 64, 64, False), (1, 5, 64, 128, 64, False), (2, 28, 28, 64, 128, False), (3, 6, 28, 64, 64, True),
    (2, 6, 16, 64, 64, False), (1, 9, 48, 64, 64, False), (2, 14, 14, 128, 64, False), (1, 14, 14, 64, 64, True),
    (1, 12, 12, 64, 64, False), (2, 8, 56, 64, 64, False), (1, 6, 24, 64, 128, True), (2, 7, 8, 64, 64, False),
    (1, 56, 56, 64, 64, True),
    # phase-form split-half wgrad of upsampled convs: low-res patches 2x16 / 1x32 / 4x8, masked rows and columns
    (2, 16, 32, 64, 64, True), (1, 16, 64, 64, 128, True), (2, 16, 16, 128, 64, True), (1, 10, 48, 64, 64, True),
    # late-fusion widths on the split-half kernel: half c-tile (C = 32) and masked k-tile (K = 32, 8)
    (2, 16, 32, 32, 32, False), (1, 28, 28, 32, 8, False), (2, 9, 7, 32, 32, False), (1, 24, 48, 32, 16, False),
    # ... and the narrow kernel (one 32 x 32 tile, waves split the 64-pixel patch 2x32 / 4x16): odd / short row counts, C < 32
    (2, 15, 32, 32, 8, False), (1, 10, 16, 8, 32, False), (1, 33, 224, 32, 32, False), (3, 3, 64, 12, 4, False),
    # ... K <= 8: the tap-packed form ((tap, k) pairs as GEMM columns, the shift on the gradient operand): 4 x 16 and 2 x 32 patches
    (1, 48, 48, 32, 8, False), (2, 17, 16, 32, 8, False), (1, 224, 224, 32, 8, False), (2, 6, 96, 20, 4, False), (1, 1, 16, 32, 8, False),
    (1, 20, 48, 32, 4, False), (2, 9, 16, 8, 4, False),        # K = 4 on the 4 x 16 patch (the entry point requires K % 4 == 0: 4 and 8 are the tap-packed widths)
])
def test_conv3x3_wgrad_split(B, Hh, Ww, C, K, ups, monkeypatch):
    """f16 x3 split-half 9-tap wgrad (ds_read_b64_tr_b16 operand transposes; dy scaled by its abs-max): patch geometries 1x32 / 2x16 / 4x8,
    masked narrow rows (28 in 32, 14 and 12 in 16), odd row counts, upsample-fused gather.  Compared with the fp64
    weight gradient; the exact-f32 kernel is held to the same bound for reference."""
    h = H()
    hin, win = (Hh // 2, Ww // 2) if ups else (Hh, Ww)
    x = rnd(B, C, hin, win, seed=21)
    dy = rnd(B, K, Hh, Ww, seed=22)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    w = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin.double(), w, None, padding=1).backward(dy.double())
    dw_bf16 = h.conv3x3_wgrad(nhwc(x), nhwc(dy), ups=ups, precision="split_bf16")
    dw_f16 = h.conv3x3_wgrad(nhwc(x), nhwc(dy), ups=ups, precision="split_f16")
    dw_f32 = h.conv3x3_wgrad(nhwc(x), nhwc(dy), ups=ups, precision="f32")
    e_bf16, e_f16, e_f32 = (rel(t.cpu().double(), w.grad) for t in (dw_bf16, dw_f16, dw_f32))
    print(f"wgrad err vs fp64: bf16x3 {e_bf16:.2e}  f16x3 {e_f16:.2e}  f32 {e_f32:.2e}")
    assert e_f32 < 1e-5
    assert e_bf16 < 2e-5
    assert e_f16 < 2e-6
    if K <= 8 and C <= 32 and Ww % 16 == 0 and not ups:          # the nine-tile form of the narrow kernel, which K <= 8 no longer takes by default
        monkeypatch.setattr(h, "WGRAD_TAPPACK", False)
        e_old = rel(h.conv3x3_wgrad(nhwc(x), nhwc(dy), precision="split_f16").cpu().double(), w.grad)
        print(f"wgrad err vs fp64, nine zero-padded column tiles: f16x3 {e_old:.2e}")
        assert e_old < 2e-6


@pytest.mark.parametrize("C", [3, 20])
@pytest.mark.parametrize("B,Hh,Ww", [(2, 16, 16), (1, 13, 21)])
def test_conv_first(C, B, Hh, Ww):
    h = H()
    x = rnd(B, C, Hh, Ww, seed=11)
    w = rnd(64, C, 3, 3, seed=12, scale=(2.0 / (9 * C)) ** 0.5).requires_grad_(True)
    b = rnd(64, seed=13, scale=0.1)
    ref = F.conv2d(x, w, b, padding=1)
    y, stat = h.conv_first_fwd(x.to(DEV), w.detach().to(DEV), b.to(DEV), True)
    assert rel(nchw(y), ref) < 2e-5
    s = stat.sum(0).cpu()
    assert rel(s[0], ref.detach().double().sum(dim=(0, 2, 3))) < 1e-5
    assert rel(s[1], (ref.detach().double() ** 2).sum(dim=(0, 2, 3))) < 1e-5
    dy = rnd(B, 64, Hh, Ww, seed=14)
    ref.backward(dy)
    dw = h.conv_first_wgrad(x.to(DEV), nhwc(dy))
    assert rel(dw.cpu(), w.grad) < 2e-5


@pytest.mark.parametrize("K", [32, 64])
@pytest.mark.parametrize("C", [1, 2, 3])
@pytest.mark.parametrize("B,Hh,Ww", [(2, 16, 16), (1, 13, 21), (3, 48, 80), (1, 224, 224)])
def test_conv_first_direct_32_filters(C, B, Hh, Ww, K):
    """The direct kernel of the first late-fusion conv (C <= 3 -> 32 filters, models/late_fusion.py:10) against fp64: output,
    BN partial sums (one row per block), with and without the statistics epilogue; ragged widths shorter than the 32-pixel step."""
    h = H()
    x = rnd(B, C, Hh, Ww, seed=21)
    w = rnd(K, C, 3, 3, seed=22, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=23, scale=0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    y, stat = h.conv_first_fwd(x.to(DEV), w.to(DEV), b.to(DEV), True)
    assert stat.shape[0] == h.LIB.egz_conv_first_stat_rows_for(B, Hh, Ww, C, K) <= 512
    assert rel(nchw(y), ref) < 2e-6
    s = stat.sum(0).cpu()
    assert rel(s[0], ref.sum(dim=(0, 2, 3))) < 1e-6 and rel(s[1], (ref ** 2).sum(dim=(0, 2, 3))) < 1e-6
    y2, none = h.conv_first_fwd(x.to(DEV), w.to(DEV), None, False)
    assert none is None and rel(nchw(y2), ref - b.double().view(1, -1, 1, 1)) < 2e-6


@pytest.mark.parametrize("B,Hh,Ww,C,K", [(2, 16, 16, 64, 64), (1, 28, 28, 128, 64), (3, 9, 7, 64, 128), (2, 32, 48, 128, 256),
                                         (1, 20, 56, 64, 96)])
def test_dgrad_with_bn_sums_wide(B, Hh, Ww, C, K, monkeypatch):
    """EPI_BNSUMS on the 64- / 128-column tiles of the streamed kernel (the encoders' conv -> BN -> ReLU -> conv pairs): the data
    gradient is bit-identical to the plain launch and the two per-channel sums match an fp64 evaluation; patch and raster
    geometries, a partial last tile."""
    h = H()
    monkeypatch.setattr(h, "SPLITK", False)      # (launches with few pixel tiles stay on the split-K form, which has no such epilogue)
    g = torch.Generator().manual_seed(B * 100 + C)
    dy = (torch.randn(B, Hh, Ww, K, generator=g) * 1e-3).to(DEV)
    w = (torch.randn(K, C, 3, 3, generator=g) * 0.05).to(DEV)
    bn_y = torch.randn(B, Hh, Ww, C, generator=g).to(DEV)
    coef = torch.stack([bn_y.mean((0, 1, 2)), 1.0 / bn_y.std((0, 1, 2)), torch.rand(C, generator=g).to(DEV) + 0.5,
                        torch.randn(C, generator=g).to(DEV) * 0.3]).contiguous()
    assert h.bnsums_ok(B, Hh, Ww, C, K, h.F16X3)
    wp, st = h.conv_weight(w, "dgrad", h.F16X3, dy, C)
    assert st
    dx, sums = h.conv3x3_dgrad_bnsums(dy, wp, C, h.F16X3, bn_y, coef)
    ref = h.conv3x3_dgrad(dy, wp, C, dtype=h.F16X3, streamed=st)
    assert torch.equal(dx, ref)
    assert sums.shape == ((B * Hh * Ww + 127) // 128, 2, C)
    d64, y64, c64 = dx.double(), bn_y.double(), coef.double()
    dz = torch.where(y64 * c64[2] + c64[3] > 0, d64, torch.zeros_like(d64))
    s = sums.sum(0)
    assert rel(s[0].cpu(), dz.sum((0, 1, 2)).cpu()) < 2e-6
    assert rel(s[1].cpu(), (dz * (y64 - c64[0]) * c64[1]).sum((0, 1, 2)).cpu()) < 2e-5
    # ... and they drive the BatchNorm backward of the layer below to the same result as its own reduce pass
    dyb, dg, db = h.bn_relu_pool_bwd(bn_y, dx, coef, False, sums=sums)
    dyb0, dg0, db0 = h.bn_relu_pool_bwd(bn_y, dx, coef, False)
    assert rel(dg.cpu(), dg0.cpu()) < 2e-5 and rel(db.cpu(), db0.cpu()) < 2e-5 and rel(dyb.cpu(), dyb0.cpu()) < 2e-5


@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("B,Hh,Ww,K", [(2, 8, 8, 64), (3, 6, 10, 512), (2, 4, 4, 8)])
def test_bn_relu_pool(pool, B, Hh, Ww, K):
    h = H()
    y = (rnd(B, K, Hh, Ww, seed=15) * 1.7 + 0.3).requires_grad_(True)
    gamma = (torch.rand(K, generator=torch.Generator().manual_seed(16)) + 0.5).requires_grad_(True)
    beta = rnd(K, seed=17, scale=0.2).requires_grad_(True)
    rm0, rv0 = rnd(K, seed=18, scale=0.1), torch.rand(K, generator=torch.Generator().manual_seed(19)) + 0.5
    rm, rv = rm0.clone(), rv0.clone()
    out = F.relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5))
    if pool:
        out = F.max_pool2d(out, 2, 2)
    dout = rnd(*out.shape, seed=20)
    out.backward(dout)
    yd = nhwc(y.detach())
    rmd, rvd = rm0.to(DEV), rv0.to(DEV)
    coef = h.bn_finalize(h.channel_stats(yd), float(B * Hh * Ww), gamma.detach().to(DEV), beta.detach().to(DEV),
                         rmd, rvd, 0.1, 1e-5)
    o = h.bn_relu_pool_fwd(yd, coef, pool)
    assert rel(nchw(o), out) < 1e-5
    assert rel(rmd.cpu(), rm) < 1e-5 and rel(rvd.cpu(), rv) < 1e-5
    dy, dg, db = h.bn_relu_pool_bwd(yd, nhwc(dout), coef, pool)
    assert rel(nchw(dy), y.grad) < 5e-5
    assert rel(dg.cpu(), gamma.grad) < 5e-5 and rel(db.cpu(), beta.grad) < 5e-5
    # eval-mode coefficients
    ce = h.bn_eval_coeffs(gamma.detach().to(DEV), beta.detach().to(DEV), rmd, rvd, 1e-5)
    oe = h.bn_relu_pool_fwd(yd, ce, False)
    refe = F.relu(F.batch_norm(y.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 1e-5))
    assert rel(nchw(oe), refe) < 1e-5


def test_pool_ties_first_wins():
    """Equal positive maxima inside a 2x2 window: gradient goes to the first in scan order (torch)."""
    h = H()
    y = torch.zeros(1, 4, 2, 2)
    y[0, :, 0, 1] = 2.0
    y[0, :, 1, 0] = 2.0
    y = y.requires_grad_(True)
    K = 4
    gamma, beta = torch.ones(K), torch.zeros(K)
    out = F.max_pool2d(F.relu(y * 1.0), 2, 2)
    out.backward(torch.ones_like(out))
    coef = torch.zeros(4, K, device=DEV)
    coef[1] = 1.0
    coef[2] = 1.0                               # mean 0, invstd 1, scale 1, shift 0
    yd = nhwc(y.detach())
    o = h.bn_relu_pool_fwd(yd, coef, True)
    assert torch.equal(nchw(o), out.detach())
    dy, _, _ = h.bn_relu_pool_bwd(yd, torch.ones(1, 1, 1, K, device=DEV), coef, True)
    dz = y.grad                                   # torch's routing: all of it at scan position (0,1)
    assert dz[0, 0, 0, 1] == 1 and dz[0, 0, 1, 0] == 0
    m1 = dz.mean(dim=(0, 2, 3), keepdim=True)
    m2 = (dz * y.detach()).mean(dim=(0, 2, 3), keepdim=True)
    assert rel(nchw(dy), dz - m1 - y.detach() * m2) < 1e-6


def test_pairmax_relu_misc():
    h = H()
    a, b = rnd(2, 5, 5, 64, seed=21), rnd(2, 5, 5, 64, seed=22)
    b[0, 0, 0, :8] = a[0, 0, 0, :8]                       # ties -> first stream wins
    y2 = torch.cat((a, b), 0).to(DEV)
    z = h.pairmax_fwd(y2)
    assert torch.equal(z.cpu(), torch.maximum(a, b))
    dz = rnd(2, 5, 5, 64, seed=23)
    dy2 = h.pairmax_bwd(y2, dz.to(DEV)).cpu()
    first = a >= b
    assert torch.equal(dy2[:2], torch.where(first, dz, torch.zeros_like(dz)))
    assert torch.equal(dy2[2:], torch.where(first, torch.zeros_like(dz), dz))
    out, g = F.relu(rnd(3, 4, 4, 32, seed=24)), rnd(3, 4, 4, 32, seed=25)
    assert torch.equal(h.relu_bwd(out.to(DEV), g.to(DEV)).cpu(), g * (out > 0))
    dyb, dbb = h.relu_bwd_bias(out.to(DEV), g.to(DEV))
    assert torch.equal(dyb.cpu(), g * (out > 0))
    assert rel(dbb.cpu(), (g * (out > 0)).sum(dim=(0, 1, 2))) < 1e-6
    x = rnd(2, 24, 5, 7, seed=26)
    assert torch.equal(h.nhwc_to_nchw(h.nchw_to_nhwc(x.to(DEV))).cpu(), x)
    assert torch.equal(h.nchw_to_nhwc(x.to(DEV)).cpu(), x.permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("C", [64, 8])
def test_head_sigmoid(C):
    h = H()
    x = rnd(2, C, 9, 11, seed=27).requires_grad_(True)
    w = rnd(1, C, 1, 1, seed=28, scale=0.2).requires_grad_(True)
    b = torch.tensor([0.1], requires_grad=True)
    ref = torch.sigmoid(F.conv2d(x, w, b))
    dout = rnd(2, 1, 9, 11, seed=29)
    ref.backward(dout)
    xd = nhwc(x.detach())
    out, logits = h.conv1x1_sigmoid_fwd(xd, w.detach().to(DEV), b.detach().to(DEV), want_logits=True)
    assert rel(out.cpu(), ref[:, 0]) < 1e-6
    dx, dw, db = h.conv1x1_sigmoid_bwd(xd, w.detach().to(DEV), out, dout[:, 0].contiguous().to(DEV))
    assert rel(nchw(dx), x.grad) < 1e-5
    assert rel(dw.cpu(), w.grad) < 1e-5 and rel(db.cpu(), b.grad) < 1e-5


@pytest.mark.parametrize("C,B,Hh,Ww", [(64, 2, 9, 11), (64, 3, 40, 56), (8, 2, 9, 11)])
def test_head_sigmoid_masked_backward(C, B, Hh, Ww):
    """Head backward with the ReLU backward of the block below folded in (egz_conv1x1_sigmoid_bwd_masked) against the two-pass
    form it replaces (egz_conv1x1_sigmoid_bwd, then egz_relu_bwd_bias) and against autograd of
    sigmoid(conv1x1(relu(y))) w.r.t. y (models/model_SP.py:28-32): dx bit-identical to the two-pass form, the bias-gradient
    sums and the weight gradient to fp32 round-off, max |dx| exact."""
    h = H()
    y = rnd(B, C, Hh, Ww, seed=31).requires_grad_(True)
    w = rnd(1, C, 1, 1, seed=32, scale=0.2).requires_grad_(True)
    b = torch.tensor([0.1], requires_grad=True)
    ref = torch.sigmoid(F.conv2d(torch.relu(y), w, b))
    dout = rnd(B, 1, Hh, Ww, seed=33)
    ref.backward(dout)
    a = nhwc(torch.relu(y.detach()))                 # the block's post-ReLU output (what ConvReLU saves)
    wd, bd, dd = w.detach().to(DEV), b.detach().to(DEV), dout[:, 0].contiguous().to(DEV)
    out, _ = h.conv1x1_sigmoid_fwd(a, wd, bd)
    dx0, dw0, db0 = h.conv1x1_sigmoid_bwd(a, wd, out, dd)
    dz0, dbias0 = h.relu_bwd_bias(a, dx0)
    dx, dw, db, stat, am = h.conv1x1_sigmoid_bwd_masked(a, wd, out, dd)
    assert torch.equal(dx, dz0)
    assert torch.equal(dw, dw0) and torch.equal(db, db0)
    dbias = h.colsum_f64(stat, C)
    assert rel(dbias, dbias0) < 1e-6
    assert rel(dbias.cpu(), nchw(dx).sum((0, 2, 3))) < 1e-5
    assert rel(nchw(dx), y.grad) < 1e-5
    assert rel(dw.cpu(), w.grad) < 1e-5 and rel(db.cpu(), b.grad) < 1e-5
    torch.cuda.synchronize()
    got = h.absmax_value(am).item()
    assert got == dx.abs().max().item()


def test_adam_matches_oracle():
    from oracle import egaze_oracle as O
    h = H()
    n = 4099                                      # exercises the non-multiple-of-4 tail
    p, g = rnd(n, seed=30), rnd(n, seed=31, scale=0.01)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = (t.clone().to(DEV) for t in (p, g, m, v))
    for step in (1, 2, 3):
        p, m, v = O.adam_step(p, g, m, v, step, 1e-3)
        h.adam_step(pd, gd, md, vd, 1e-3, 0.9, 0.999, 1e-8, step)
    assert rel(pd.cpu(), p) < 1e-6 and rel(md.cpu(), m) < 1e-6 and rel(vd.cpu(), v) < 1e-6


def test_adam_skips_nonfinite_gradients():
    """a NaN / inf gradient element must not reach the weights.  The kernel leaves p, m, v of such an element alone,
    sets bit 0 of the device word and updates every other element with exactly the bits of a clean step; FusedAdam.check_finite
    raises at the next look (both counter forms: host scalar and device counter)."""
    h = H()
    n = 4099
    g = rnd(n, seed=41, scale=0.01)
    bad = g.clone()
    bad[5], bad[1030], bad[n - 1] = float("nan"), float("inf"), float("-inf")        # vector body, another block, scalar tail
    for dev_counter in (False, True):
        res = {}
        for tag, grad in (("clean", g), ("bad", bad)):
            pd, md, vd = rnd(n, seed=40).to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
            flag = torch.zeros(1, dtype=torch.int32, device=DEV)
            cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
            for step in (1, 2):
                if dev_counter:
                    h.adam_step_dev(pd, grad.to(DEV), md, vd, 1e-3, 0.9, 0.999, 1e-8, cnt, nonfinite=flag)
                else:
                    h.adam_step(pd, grad.to(DEV), md, vd, 1e-3, 0.9, 0.999, 1e-8, step, nonfinite=flag)
            res[tag] = (pd.cpu(), md.cpu(), vd.cpu(), int(flag.item()))
        assert res["clean"][3] == 0 and res["bad"][3] == 1
        keep = torch.ones(n, dtype=torch.bool)
        keep[[5, 1030, n - 1]] = False
        p0 = rnd(n, seed=40)
        for a, b in zip(res["clean"][:3], res["bad"][:3]):
            assert torch.equal(a[keep], b[keep])                       # untouched elsewhere, bit for bit
        assert torch.equal(res["bad"][0][~keep], p0[~keep])            # the poisoned elements kept their parameters ...
        assert res["bad"][1][~keep].abs().max().item() == 0.0 and res["bad"][2][~keep].abs().max().item() == 0.0    # ... and moments
        assert torch.isfinite(res["bad"][0]).all()
    from egaze_amd.optim import FusedAdam
    w = torch.nn.Parameter(rnd(64, seed=42).to(DEV))
    opt = FusedAdam([w], lr=1e-3)
    opt.zero_grad()
    w.grad.copy_(rnd(64, seed=43).to(DEV))
    opt.step()
    opt.check_finite()
    before = w.detach().clone()
    w.grad[3] = float("nan")
    opt.step()
    with pytest.raises(FloatingPointError):
        opt.check_finite()
    opt.check_finite()                                                 # the flag was reset by the failed check
    assert torch.isfinite(w).all() and w[3].item() == before[3].item()


def test_mse():
    h = H()
    a, b = rnd(1, 1, 512, seed=32).requires_grad_(True), rnd(1, 1, 512, seed=33)
    ref = ((a - b) ** 2).mean()
    ref.backward()
    ad, bd = a.detach().to(DEV), b.to(DEV)
    assert abs(h.mse_fwd(ad, bd).item() - ref.item()) < 1e-6 * abs(ref.item())
    assert rel(h.mse_bwd(ad, bd, None).cpu(), a.grad) < 1e-6
    # criterion(pred, tanh(target)) of AT.py:138 as one kernel, through the autograd node
    from egaze_amd.functions import MSELoss
    a2 = a.detach().clone().requires_grad_(True)
    ref2 = ((a2 - torch.tanh(b)) ** 2).mean()
    ref2.backward()
    a3 = a.detach().to(DEV).requires_grad_(True)
    l3 = MSELoss.apply(a3, bd, True)
    l3.backward(gradient=torch.ones((), device=DEV))
    assert abs(l3.item() - ref2.item()) < 2e-6 * abs(ref2.item()) and rel(a3.grad.cpu(), a2.grad) < 2e-6
    # the stream-ordered copy / zero helpers are plain kernels for aligned buffers, the runtime calls otherwise
    src = torch.randn(1031, device=DEV)
    dst = torch.full((1031,), 7.0, device=DEV)
    h.copy_into(dst[:1028], src[:1028])
    assert torch.equal(dst[:1028], src[:1028]) and float(dst[1028]) == 7.0
    h.copy_into(dst[1:1030], src[2:1031])                    # unaligned: hipMemcpyAsync route
    assert torch.equal(dst[1:1030], src[2:1031])
    h.fill_zero(dst[:512])
    assert float(dst[:512].abs().max()) == 0.0 and float(dst[512]) == float(src[513])
    h.fill_zero(dst[513:])                                   # unaligned start
    assert float(dst[513:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype,tol", [(1, 2e-6), (2, 3e-5)])
@pytest.mark.parametrize("B,Hh,Ww,C,K", [(2, 12, 12, 64, 128), (1, 14, 14, 256, 256), (3, 9, 7, 32, 128),
                                        (2, 10, 12, 64, 64), (1, 8, 8, 128, 64),
                                        # halo-tile kernel geometries: 8 x 16 patches (W % 16 == 0, H % 8 == 0) ...
                                        (2, 16, 32, 64, 128), (1, 8, 16, 64, 64), (3, 24, 48, 32, 128),
                                        # ... and raster runs crossing row ends / image borders (28, 56 wide; ragged M)
                                        (3, 28, 28, 64, 128), (1, 20, 56, 96, 64), (5, 5, 3, 64, 64)])
def test_conv3x3_split_half(B, Hh, Ww, C, K, dtype, tol):
    """Error-compensated split-half operands (f16 x3 / bf16 x3 on the 16-bit MFMA path) against an fp64 reference:
    f16 x3 must be fp32-class (the exact-f32 kernel itself sits at ~5e-7), bf16 x3 within 3e-5."""
    h = H()
    x = rnd(B, C, Hh, Ww, seed=41)
    w = rnd(K, C, 3, 3, seed=42, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=43, scale=0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    wd = w.to(DEV)
    for epi in (0, 1, 2):
        want = F.relu(ref) if epi == 1 else ref
        y, stat = h.conv3x3_fwd(nhwc(x), h.packed_weight(wd, "fwd", dtype), b.to(DEV), K, epi=epi, dtype=dtype)
        assert rel(nchw(y), want) < tol
        if epi == 2:
            assert rel(stat.sum(0)[0].cpu(), ref.sum(dim=(0, 2, 3))) < 1e-5
    yg, sg = h.conv3x3_fwd(nhwc(x), h.packed_weight(wd, "fwd", dtype), b.to(DEV), K, epi=2, dtype=dtype, tile_flag=0x2000)
    assert rel(y, yg) < 2e-6 and rel(stat.sum(0), sg.sum(0)) < 1e-6       # halo-tile vs per-tap gather kernel
    y32, _ = h.conv3x3_fwd(nhwc(x), h.packed_weight(wd, "fwd"), b.to(DEV), K, epi=0)
    print(f"dtype {dtype}: split err {rel(nchw(y), ref):.2e}, exact-f32 MFMA err {rel(nchw(y32), ref):.2e}")
    # data gradient (GEMM output channels = C must be a multiple of 128 for the split kernel)
    if C % 64 == 0:
        dy = rnd(B, K, Hh, Ww, seed=44)
        dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
        dx = h.conv3x3_dgrad(nhwc(dy), h.packed_weight(wd, "dgrad", dtype), C, dtype=dtype)
        assert rel(nchw(dx), dref) < tol


@pytest.mark.parametrize("B,Hh,Ww,C,K,ups", [
    (1, 148, 224, 32, 256, False),     # 518 tiles = one full round of 512 + 6 tail tiles split 9 ways
    (1, 64, 100, 64, 128, False),      # 50 tiles < one round: every tile split 9 ways
    (2, 56, 58, 32, 64, True),         # phase-decomposed upsample, 4 x 26 tiles of 128 x 64
])
def test_conv3x3_split_tail_schedule(B, Hh, Ww, C, K, ups):
    """Split-K tail of the split-half implicit GEMM (main round + split tail + fixed-order fixup) against the plain
    single launch of the same kernel (the default; the schedule is opt-in, flag 0x4000): same values up to fp32
    summation order, same BN statistics."""
    h = H()
    x = rnd(B, C, Hh // 2 if ups else Hh, Ww // 2 if ups else Ww, seed=51)
    w = rnd(K, C, 3, 3, seed=52, scale=(2.0 / (9 * C)) ** 0.5).to(DEV)
    b = rnd(K, seed=53, scale=0.1).to(DEV)
    xd = nhwc(x)
    for dtype in (1, 2):
        wp = h.packed_weight(w, "ups_fwd" if ups else "fwd", dtype)
        for epi in (1, 2):
            y0, s0 = h.conv3x3_fwd(xd, wp, b, K, ups="phase" if ups else False, epi=epi, dtype=dtype)
            y1, s1 = h.conv3x3_fwd(xd, wp, b, K, ups="phase" if ups else False, epi=epi, dtype=dtype, tile_flag=0x4000)
            assert rel(y1, y0) < 2e-6
            if epi == 2:
                assert s0.shape == s1.shape
                assert rel(s1.sum(0), s0.sum(0)) < 1e-6
    # the transposed (data-gradient) role on the same schedule
    dy = nhwc(rnd(B, K, Hh, Ww, seed=54))
    if C % 64 == 0 and not ups:
        wd = h.packed_weight(w, "dgrad", 2)
        d0, _ = h.conv3x3_fwd(dy, wd, None, C, epi=0, dtype=2)
        d1, _ = h.conv3x3_fwd(dy, wd, None, C, epi=0, dtype=2, tile_flag=0x4000)
        assert rel(d1, d0) < 2e-6


@pytest.mark.parametrize("dtype,tol", [(1, 2e-6), (2, 3e-5)])
def test_conv3x3_split_half_upsample(dtype, tol):
    h = H()
    B, Hh, Ww, C, K = 2, 6, 8, 128, 128
    x = rnd(B, C, Hh, Ww, seed=45).requires_grad_(True)
    w = rnd(K, C, 3, 3, seed=46, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=47, scale=0.1)
    xd, wd = x.detach().double().requires_grad_(True), w.double()
    pre = F.conv2d(F.interpolate(xd, scale_factor=2, mode="nearest"), wd, b.double(), padding=1)
    dy = rnd(B, K, 2 * Hh, 2 * Ww, seed=48)
    pre.backward(dy.double())
    wdev = w.to(DEV)
    y, _ = h.conv3x3_fwd(nhwc(x.detach()), h.packed_weight(wdev, "ups_fwd", dtype), b.to(DEV), K, ups="phase", epi=0, dtype=dtype)
    assert rel(nchw(y), pre.detach()) < tol
    dx = h.conv3x3_ups_dgrad(nhwc(dy), h.packed_weight(wdev, "ups_dgrad", dtype), C, dtype=dtype)
    assert rel(nchw(dx), xd.grad) < tol


def test_split_kernels_shape_fuzz():
    """Geometry dispatch fuzz: random (B, H, W, C, K) through every split-half kernel family (halo patch / halo raster run /
    per-tap gather forward and data gradient, 9-tap and phase-form weight gradients, upsample forms) against the exact-f32
    kernels of the same library.  Catches holes in the patch / run / mask selection rather than arithmetic."""
    h = H()
    rs = np.random.RandomState(2024)
    chans = [32, 64, 96, 128, 192, 256]
    for it in range(36):
        B = int(rs.randint(1, 4))
        Hh = int(rs.choice([2, 3, 4, 6, 7, 8, 9, 14, 16, 24, 28, 30, 32, 40, 56, 57]))
        Ww = int(rs.choice([2, 3, 4, 6, 7, 8, 12, 14, 16, 24, 28, 32, 48, 56, 60, 64, 80]))
        C, K = int(rs.choice(chans)), int(rs.choice([64, 128, 192, 256]))
        ups = bool(it % 3 == 0)
        x = torch.from_numpy(rs.standard_normal((B, Hh, Ww, C)).astype(np.float32)).to(DEV)
        w = torch.from_numpy((rs.standard_normal((K, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32)).to(DEV)
        b = torch.from_numpy(rs.standard_normal(K).astype(np.float32) * 0.1).to(DEV)
        tag = (it, B, Hh, Ww, C, K, ups)
        mode = "phase" if ups else False
        Ho, Wo = (2 * Hh, 2 * Ww) if ups else (Hh, Ww)
        y0, s0 = h.conv3x3_fwd(x, h.packed_weight(w, "ups_fwd" if ups else "fwd", 0), b, K, ups=mode, epi=2, dtype=0)
        y1, s1 = h.conv3x3_fwd(x, h.packed_weight(w, "ups_fwd" if ups else "fwd", 1), b, K, ups=mode, epi=2, dtype=1)
        assert rel(y1, y0) < 5e-6, tag
        assert rel(s1.sum(0), s0.sum(0)) < 1e-5, tag
        dy = torch.from_numpy(rs.standard_normal((B, Ho, Wo, K)).astype(np.float32)).to(DEV)
        if C % 64 == 0:
            if ups:
                d0 = h.conv3x3_ups_dgrad(dy, h.packed_weight(w, "ups_dgrad", 0), C, dtype=0)
                d1 = h.conv3x3_ups_dgrad(dy, h.packed_weight(w, "ups_dgrad", 2), C, dtype=2)
            else:
                d0 = h.conv3x3_dgrad(dy, h.packed_weight(w, "dgrad", 0), C, dtype=0)
                d1 = h.conv3x3_dgrad(dy, h.packed_weight(w, "dgrad", 2), C, dtype=2)
                d2 = h.conv3x3_dgrad(dy, h.packed_weight(w, "dgrad", 1), C, dtype=1)      # abs-max scaled f16 x3
                assert rel(d2, d0) < 5e-6, tag
            assert rel(d1, d0) < 4e-5, tag
            g0 = h.conv3x3_wgrad(x, dy, ups=ups, precision="f32")
            g1 = h.conv3x3_wgrad(x, dy, ups=ups, precision="split_f16")
            assert rel(g1, g0) < 5e-6, tag
            g2 = h.conv3x3_wgrad(x, dy, ups=ups, precision="split_bf16")
            assert rel(g2, g0) < 4e-5, tag


@pytest.mark.parametrize("mag", [1.0, 3e-4, 1e-8, 7e-13, 2e5])
def test_gradient_absmax_scaling(mag):
    """f16 x3 data / weight gradients of a gradient tensor of ANY magnitude: the abs-max of dy (a producer's or
    egz_absmax) picks a power-of-two scale, so 1e-8-sized gradients keep fp32-class accuracy (unscaled f16 would flush
    them to zero, bf16 x3 carries 16 bits)."""
    h = H()
    B, Hh, Ww, C, K = 2, 16, 32, 64, 128
    x = rnd(B, C, Hh, Ww, seed=81)
    w = rnd(K, C, 3, 3, seed=82, scale=(2.0 / (9 * C)) ** 0.5)
    dy = rnd(B, K, Hh, Ww, seed=83) * mag
    dy[0, 3, 5, 7] = 40.0 * mag                                      # an outlier sets the scale
    dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    dyd, wd = nhwc(dy), w.to(DEV)
    am = h.absmax_of(dyd)
    assert torch.equal(h.absmax_value(am).cpu(), dy.abs().max().reshape(1))     # bit pattern of the exact max
    dx = h.conv3x3_dgrad(dyd, h.packed_weight(wd, "dgrad", 1), C, dtype=1)
    assert rel(nchw(dx), dref) < 2e-6
    dw = h.conv3x3_wgrad(nhwc(x), dyd, precision="split_f16")
    assert rel(dw.cpu(), wref) < 2e-6
    assert rel(h.conv3x3_wgrad(nhwc(x), dyd, precision="split_bf16").cpu(), wref) < 2e-5
    # a producer attaches the same scalar in its own pass
    out = torch.rand(B, Hh, Ww, K, device=DEV) - 0.3
    keep = (h.PRECISION, h.GRAD_SPLIT)
    h.PRECISION, h.GRAD_SPLIT = "split", "f16"
    try:
        dy2, _ = h.relu_bwd_bias(out, dyd)
    finally:
        h.PRECISION, h.GRAD_SPLIT = keep
    want = (dy2.abs().max()).reshape(1).cpu()
    assert torch.equal(h.absmax_value(dy2._egz_absmax).cpu(), want)


@pytest.mark.parametrize("mag", [1e-5, 1e-2, 1.0, 1e3, 2e5])
def test_forward_activation_scaling(mag):
    """f16 x3 forward operands of ANY magnitude (round-3 parity item): the pass that writes a post-ReLU activation also
    emits max |a| (BN apply + ReLU [+ pool]; the bias + ReLU epilogue of the streamed / phase-upsample conv kernels) and the
    consuming convolution -- forward operand, weight-gradient x operand -- scales by the matching power of two before the
    split.  Unscaled, activations above 65504 overflow the hi half and below ~0.125 push the lo half into f16 subnormals;
    the reference's fp32 Conv2d (utils.py:70, models/model_SP.py:13-29) has no such domain.  Bound: 2e-6 of max |ref| vs fp64,
    the same bound as at magnitude 1."""
    h = H()
    assert h.FWD_SCALE
    keep = h.PRECISION
    h.PRECISION = "split"
    try:
        B, Hh, Ww, C, K = 2, 16, 32, 64, 128
        # ---- producer 1: BN apply + ReLU + pool writes the activation and its abs-max in one pass
        ypre = rnd(B, 2 * Hh, 2 * Ww, C, seed=71).to(DEV)
        coef = torch.zeros(4, C, device=DEV)
        coef[2] = mag
        coef[3] = 0.1 * mag
        a = h.bn_relu_pool_fwd(ypre, coef, True)
        am = a._egz_absmax
        assert torch.equal(h.absmax_value(am).cpu(), a.max().reshape(1).cpu())          # exact max, as a bit pattern
        a_ref = a.permute(0, 3, 1, 2).double().cpu()
        w = rnd(K, C, 3, 3, seed=72, scale=(2.0 / (9 * C)) ** 0.5)
        b = rnd(K, seed=73, scale=0.1 * mag)
        wd = torch.nn.Parameter(w.to(DEV))
        ref = F.relu(F.conv2d(a_ref, w.double(), b.double(), padding=1))
        before = h.ABSMAX_STATS["standalone"]
        wq, st = h.conv_weight(wd, "fwd", h.F16X3, a, K)
        assert st
        y, _ = h.conv3x3_fwd(a, wq, b.to(DEV), K, epi=h.EPI_BIAS_RELU, dtype=h.F16X3, streamed=True)
        assert h.ABSMAX_STATS["standalone"] == before          # the producer's scalar was used, no extra pass
        assert rel(nchw(y), ref) < 2e-6
        # ---- producer 2: the conv's own bias + ReLU epilogue emitted max |y| for the next layer
        assert torch.equal(h.absmax_value(y._egz_absmax).cpu(), y.max().reshape(1).cpu())
        # ---- the gather-kernel family (phase-upsample forward) scales the same way and emits its abs-max too
        w2 = rnd(64, K, 3, 3, seed=74, scale=(2.0 / (9 * K)) ** 0.5)
        w2d = torch.nn.Parameter(w2.to(DEV))
        y_ref = nchw(y).double()
        ref2 = F.relu(F.conv2d(F.interpolate(y_ref, scale_factor=2, mode="nearest"), w2.double(), None, padding=1))
        y2, _ = h.conv3x3_fwd(y, h.packed_weight(w2d, "ups_fwd", h.F16X3), None, 64, ups="phase", epi=h.EPI_BIAS_RELU,
                              dtype=h.F16X3)
        assert h.ABSMAX_STATS["standalone"] == before
        assert rel(nchw(y2), ref2) < 2e-6
        assert torch.equal(h.absmax_value(y2._egz_absmax).cpu(), y2.max().reshape(1).cpu())
        # ---- weight gradient: the x operand is scaled by its abs-max, dy by its own
        dy = rnd(B, K, Hh, Ww, seed=75)
        wref = torch.nn.grad.conv2d_weight(a_ref, w.shape, dy.double(), padding=1)
        dw = h.conv3x3_wgrad(a, nhwc(dy), precision="split_f16")
        assert rel(dw.cpu(), wref) < 2e-6
        wref2 = torch.nn.grad.conv2d_weight(F.interpolate(y_ref, scale_factor=2, mode="nearest"), w2.shape,
                                            rnd(B, 64, 2 * Hh, 2 * Ww, seed=76).double(), padding=1)
        dw2 = h.conv3x3_wgrad(y, nhwc(rnd(B, 64, 2 * Hh, 2 * Ww, seed=76)), ups=True, precision="split_f16")
        assert rel(dw2.cpu(), wref2) < 2e-6
        # ---- an operand without a producer scalar (network input, user tensor): one standalone pass, same bound
        a2 = a.clone()
        before = h.ABSMAX_STATS["standalone"]
        y3, _ = h.conv3x3_fwd(a2, wq, b.to(DEV), K, epi=h.EPI_BIAS, dtype=h.F16X3, streamed=True)
        assert h.ABSMAX_STATS["standalone"] == before + 1
        assert rel(nchw(y3), F.conv2d(a_ref, w.double(), b.double(), padding=1)) < 2e-6
    finally:
        h.PRECISION = keep


def test_forward_scaling_off_loses_the_small_range(monkeypatch):
    """The knob that switches the forward scaling off (hipops.FWD_SCALE = False) shows what it is for: at max |a| = 1e-5
    the unscaled f16 pair is two orders of magnitude less accurate than the scaled one."""
    h = H()
    B, Hh, Ww, C, K = 1, 16, 16, 64, 128
    a = (torch.rand(B, Hh, Ww, C, generator=torch.Generator().manual_seed(5)) * 1e-5).to(DEV)
    w = rnd(K, C, 3, 3, seed=6, scale=(2.0 / (9 * C)) ** 0.5)
    wd = torch.nn.Parameter(w.to(DEV))
    ref = F.conv2d(a.permute(0, 3, 1, 2).double().cpu(), w.double(), None, padding=1)
    wq, st = h.conv_weight(wd, "fwd", h.F16X3, a, K)
    y_on, _ = h.conv3x3_fwd(a, wq, None, K, epi=h.EPI_BIAS, dtype=h.F16X3, streamed=st)
    monkeypatch.setattr(h, "FWD_SCALE", False)
    y_off, _ = h.conv3x3_fwd(a.clone(), wq, None, K, epi=h.EPI_BIAS, dtype=h.F16X3, streamed=st)
    e_on, e_off = rel(nchw(y_on), ref), rel(nchw(y_off), ref)
    assert e_on < 2e-6 and e_off > 20 * e_on, (e_on, e_off)


@pytest.mark.parametrize("B,Hh,Ww,C", [(2, 16, 32, 20), (1, 24, 28, 20), (2, 9, 7, 17)])
def test_first_conv_padded_split_path(B, Hh, Ww, C):
    """The flow-stack first conv (Cin = 20, SP.py:53) on the split-half kernels after zero-padding Cin to 32:
    padded transpose, forward with BN statistics, and the C = 32 half-tile weight gradient."""
    h = H()
    K = 64
    x = rnd(B, C, Hh, Ww, seed=91)
    w = rnd(K, C, 3, 3, seed=92, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=93, scale=0.1)
    dy = rnd(B, K, Hh, Ww, seed=94)
    xp = h.nchw_to_nhwc_pad(x.to(DEV), 32)
    assert tuple(xp.shape) == (B, Hh, Ww, 32)
    assert torch.equal(xp[..., :C].cpu(), x.permute(0, 2, 3, 1)) and float(xp[..., C:].abs().max()) == 0.0
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    wd = torch.nn.Parameter(w.to(DEV))
    y, stat = h.conv3x3_fwd(xp, h.packed_weight(wd, "fwd", 1), b.to(DEV), K, epi=2, dtype=1)
    assert rel(nchw(y), ref) < 2e-6
    assert rel(stat.sum(0)[0].cpu(), ref.sum(dim=(0, 2, 3))) < 1e-5
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    for prec, tol in (("split_bf16", 2e-5), ("split_f16", 2e-6), ("f32", 1e-5)):
        dw = h.conv3x3_wgrad(xp, nhwc(dy), precision=prec)
        assert tuple(dw.shape) == (K, 32, 3, 3)
        assert rel(dw[:, :C].cpu(), wref) < tol, prec
        assert float(dw[:, C:].abs().max()) == 0.0, prec


# ----------------------------------------------------------------------------- streamed-weight kernel (conv3x3_igemm_x3s)
@pytest.mark.parametrize("dtype,tol", [(1, 2e-6), (2, 3e-5)])
@pytest.mark.parametrize("B,Hh,Ww,C,K", [
    # patch geometry, 128-column tiles (8 x 16 patches) / 64-column tiles (16 x 16 patches)
    (2, 16, 32, 64, 128), (1, 8, 16, 32, 256), (3, 24, 48, 96, 128), (2, 16, 16, 64, 64), (1, 32, 48, 32, 64),
    # raster runs: row ends, image borders, ragged last tile, several channel blocks (double-buffered image switches)
    (3, 28, 28, 64, 128), (2, 14, 14, 256, 256), (1, 56, 56, 64, 128), (5, 5, 3, 64, 64), (1, 20, 56, 96, 64),
    (2, 12, 12, 128, 64), (3, 9, 7, 32, 128), (1, 8, 24, 64, 192),
    # 32-column tiles (late-fusion widths): K = 32 / 8 (zero-padded columns), C = 8 (one zero-padded channel block)
    (2, 16, 16, 32, 32), (1, 20, 56, 32, 8), (2, 12, 12, 8, 32), (1, 32, 48, 32, 32), (3, 28, 28, 16, 40),
    # ... of which images that are multiples of 16 with C, K <= 32 run on the persistent narrow kernel (weights resident in
    # registers, halo of the next tile prefetched): several images (border masks between them), K = 8, C = 8, and more
    # tiles than blocks (320 tiles on 256 CUs: the per-XCD tile ranges and the double-buffered image switch)
    (3, 32, 32, 32, 8), (2, 48, 16, 8, 32), (5, 128, 128, 32, 32), (9, 16, 16, 12, 20)])
def test_conv3x3_streamed(B, Hh, Ww, C, K, dtype, tol):
    """Streamed-weight halo kernel (fragment-ordered weights L2 -> registers, activation halo through LDS) against an fp64
    reference: forward with all three epilogues (BN partial sums included) and the data gradient, f16 x3 and bf16 x3."""
    h = H()
    x = rnd(B, C, Hh, Ww, seed=61)
    w = rnd(K, C, 3, 3, seed=62, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=63, scale=0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    wd, xd = w.to(DEV), nhwc(x)
    wp, st = h.conv_weight(wd, "fwd", dtype, xd, K)
    assert st, "geometry expected on the streamed kernel"
    for epi in (0, 1, 2):
        want = F.relu(ref) if epi == 1 else ref
        y, stat = h.conv3x3_fwd(xd, wp, b.to(DEV), K, epi=epi, dtype=dtype, streamed=True)
        assert rel(nchw(y), want) < tol, epi
        if epi == 2:      # one partial row per 128 pixels (per 32 when the launch ran split-K: few pixel tiles), or one per
            # block of the persistent narrow kernel (one block per CU at most, a multiple of 8)
            per = 32 if h.LIB.egz_conv3x3_streamed_splits(B, Hh, Ww, C, K) > 1 else 128
            if C <= 32 and K <= 32 and Hh % 16 == 0 and Ww % 16 == 0:
                assert stat.shape[0] == h.LIB.egz_conv3x3_streamed_stat_rows(B, Hh, Ww, C, K) <= 256 and stat.shape[0] % 8 == 0
            else:
                assert stat.shape[0] == (B * Hh * Ww + per - 1) // per
            assert rel(stat.sum(0)[0].cpu(), ref.sum(dim=(0, 2, 3))) < 1e-5
            assert rel(stat.sum(0)[1].cpu(), (ref * ref).sum(dim=(0, 2, 3))) < 1e-5
    if C % 32 == 0 and K % 64 == 0:      # same values as the LDS-DMA halo kernel up to fp32 summation order
        yh, sh = h.conv3x3_fwd(xd, h.packed_weight(wd, "fwd", dtype), b.to(DEV), K, epi=2, dtype=dtype)
        assert rel(y, yh) < 2e-6 and rel(stat.sum(0), sh.sum(0)) < 1e-6
    if (C % 64 == 0 or C <= 40) and (K % 32 == 0 or (K < 32 and K % 4 == 0)):      # reduction over K must be streamable
        dy = rnd(B, K, Hh, Ww, seed=64)
        dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
        dyd = nhwc(dy)
        wq, st = h.conv_weight(wd, "dgrad", dtype, dyd, C)
        assert st
        dx = h.conv3x3_dgrad(dyd, wq, C, dtype=dtype, streamed=True)
        assert rel(nchw(dx), dref) < tol


@pytest.mark.parametrize("B,Hh,Ww,C,K", [(1, 28, 28, 512, 512), (1, 14, 14, 256, 512), (2, 16, 16, 128, 128),
                                         (1, 56, 56, 128, 256), (3, 14, 14, 512, 128), (1, 7, 9, 160, 128)])
def test_conv3x3_streamed_splitk(B, Hh, Ww, C, K, monkeypatch):
    """Split-K form of the streamed kernel (few pixel tiles: batch-1 inference, 14 x 14 layers): the channel blocks of a tile
    are divided over several blocks and a fix-up pass sums them and applies the epilogue.  All three epilogues and the data
    gradient against fp64 and against the unsplit launch (same values up to fp32 summation order, same BN partial sums)."""
    h = H()
    ns = h.LIB.egz_conv3x3_streamed_splits(B, Hh, Ww, C, K)
    assert ns >= 2, "geometry expected to be split"
    x = rnd(B, C, Hh, Ww, seed=91)
    w = rnd(K, C, 3, 3, seed=92, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=93, scale=0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    wd, xd = w.to(DEV), nhwc(x)
    wp, st = h.conv_weight(wd, "fwd", 1, xd, K)
    assert st
    for epi in (0, 1, 2):
        want = F.relu(ref) if epi == 1 else ref
        monkeypatch.setattr(h, "SPLITK", True)
        y, stat = h.conv3x3_fwd(xd, wp, b.to(DEV), K, epi=epi, dtype=1, streamed=True)
        monkeypatch.setattr(h, "SPLITK", False)
        y0, stat0 = h.conv3x3_fwd(xd, wp, b.to(DEV), K, epi=epi, dtype=1, streamed=True)
        assert rel(nchw(y), want) < 2e-6, epi
        assert rel(y, y0) < 2e-6, epi
        if epi == 2:
            assert stat.shape[0] == (B * Hh * Ww + 31) // 32 and stat0.shape[0] == (B * Hh * Ww + 127) // 128
            assert rel(stat.sum(0), stat0.sum(0)) < 1e-6
            assert rel(stat.sum(0)[0].cpu(), ref.sum(dim=(0, 2, 3))) < 1e-5
            assert rel(stat.sum(0)[1].cpu(), (ref * ref).sum(dim=(0, 2, 3))) < 1e-5
    if K % 32 == 0 and C % 128 == 0 and h.LIB.egz_conv3x3_streamed_splits(B, Hh, Ww, K, C) >= 2:
        monkeypatch.setattr(h, "SPLITK", True)
        dy = rnd(B, K, Hh, Ww, seed=94, scale=1e-4)                 # small gradients: the abs-max scaling path
        dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
        dyd = nhwc(dy)
        wq, st = h.conv_weight(wd, "dgrad", 1, dyd, C)
        assert st
        dx = h.conv3x3_dgrad(dyd, wq, C, dtype=1, streamed=True)
        assert rel(nchw(dx), dref) < 2e-6


def test_conv3x3_streamed_shape_fuzz():
    """Random geometries through the streamed kernel wherever egz_conv3x3_streamed_ok accepts them, against the exact-f32
    kernels: patch / run selection, ragged tiles, XCD tile map with tile counts that are not multiples of 8."""
    h = H()
    rs = np.random.RandomState(77)
    n_streamed = 0
    for it in range(40):
        B = int(rs.randint(1, 5))
        Hh = int(rs.choice([2, 3, 5, 7, 8, 14, 16, 24, 28, 30, 32, 48, 56, 57]))
        Ww = int(rs.choice([2, 3, 4, 7, 8, 14, 16, 28, 32, 48, 56, 60, 62, 64, 80]))
        C, K = int(rs.choice([32, 64, 96, 128, 256])), int(rs.choice([64, 128, 192, 256]))
        x = torch.from_numpy(rs.standard_normal((B, Hh, Ww, C)).astype(np.float32)).to(DEV)
        w = torch.from_numpy((rs.standard_normal((K, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32)).to(DEV)
        b = torch.from_numpy(rs.standard_normal(K).astype(np.float32) * 0.1).to(DEV)
        wp, st = h.conv_weight(w, "fwd", 1, x, K)
        if not st:
            assert Ww > 62 and (Ww % 16 != 0 or Hh % (8 if K % 128 == 0 else 16) != 0), (it, B, Hh, Ww, C, K)
            continue
        n_streamed += 1
        y0, s0 = h.conv3x3_fwd(x, h.packed_weight(w, "fwd", 0), b, K, epi=2, dtype=0)
        y1, s1 = h.conv3x3_fwd(x, wp, b, K, epi=2, dtype=1, streamed=True)
        assert rel(y1, y0) < 5e-6, (it, B, Hh, Ww, C, K)
        assert rel(s1.sum(0), s0.sum(0)) < 1e-5, (it, B, Hh, Ww, C, K)
    assert n_streamed >= 25


@pytest.mark.parametrize("dtype,tol", [(1, 2e-6), (2, 3e-5)])
@pytest.mark.parametrize("B,Hl,Wl,C,K", [
    (2, 16, 32, 128, 64),      # low-res patch geometry (hi-res 32 x 64)
    (1, 8, 16, 256, 32),
    (3, 28, 28, 128, 64),      # raster runs on the polyphase components, several channel blocks
    (2, 14, 14, 128, 96), (1, 5, 3, 128, 64), (2, 9, 7, 256, 128)])
def test_conv3x3_streamed_ups_dgrad(B, Hl, Wl, C, K, dtype, tol):
    """Streamed polyphase form of the data gradient of [nearest x2 upsample -> conv3x3] w.r.t. the low-res input
    (models/model_SP.py:17-18 etc.) against autograd in fp64 and against the per-tap gather kernel."""
    h = H()
    x = rnd(B, C, Hl, Wl, seed=71).double().requires_grad_(True)
    w = rnd(K, C, 3, 3, seed=72, scale=(2.0 / (9 * C)) ** 0.5)
    dy = rnd(B, K, 2 * Hl, 2 * Wl, seed=73)
    F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w.double(), None, padding=1).backward(dy.double())
    wd, dyd = w.to(DEV), nhwc(dy)
    wq, st = h.conv_weight(wd, "ups_dgrad", dtype, dyd, C)
    assert st, "geometry expected on the streamed kernel"
    dx = h.conv3x3_ups_dgrad(dyd, wq, C, dtype=dtype, streamed=True)
    assert rel(nchw(dx), x.grad) < tol
    dg = h.conv3x3_ups_dgrad(dyd, h.packed_weight(wd, "ups_dgrad", dtype), C, dtype=dtype)
    assert rel(dx, dg) < (2e-6 if dtype == 1 else 3e-5)


@pytest.mark.parametrize("epi_relu", [True, False])
@pytest.mark.parametrize("B,Hl,Wl,C,K", [
    (2, 16, 32, 128, 64),      # 64-column tile, low-res patch geometry (the 128 -> 64 @ 224 layer's form)
    (2, 16, 16, 64, 128),      # 128-column tile, patch geometry
    (3, 14, 14, 128, 128),     # raster runs, several channel blocks (the 28 x 28 / 56 x 56 layers' form)
    (2, 28, 28, 64, 256),      # two column tiles
    (1, 5, 3, 64, 64), (2, 9, 7, 96, 192), (1, 12, 12, 32, 64)])
def test_conv3x3_streamed_ups_fwd(B, Hl, Wl, C, K, epi_relu):
    """Forward of [nearest x2 upsample -> conv3x3 (-> ReLU)] (models/model_SP.py:16-18 etc.) as four phase convolutions on the
    streamed-weight kernel (MODE UPSF, kind-7 packing) against fp64 torch and against the per-tap gather kernel it replaces;
    the max |y| the bias + ReLU epilogue emits for the next layer's f16 split must be exact."""
    h = H()
    x = rnd(B, C, Hl, Wl, seed=81)
    w = rnd(K, C, 3, 3, seed=82, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=83, scale=0.1)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    if epi_relu:
        ref = torch.relu(ref)
    xd, wd, bd = nhwc(x), w.to(DEV), b.to(DEV)
    epi = h.EPI_BIAS_RELU if epi_relu else h.EPI_BIAS
    wq, st = h.conv_weight(wd, "ups_fwd", h.F16X3, xd, K)
    assert st, "geometry expected on the streamed kernel"
    y, _ = h.conv3x3_fwd(xd, wq, bd, K, ups="phase", epi=epi, dtype=h.F16X3, streamed=True)
    assert tuple(y.shape) == (B, 2 * Hl, 2 * Wl, K)
    assert rel(nchw(y), ref) < 2e-6
    y0, _ = h.conv3x3_fwd(xd, h.packed_weight(wd, "ups_fwd", h.F16X3), bd, K, ups="phase", epi=epi, dtype=h.F16X3)
    assert rel(y, y0) < 2e-6
    if epi_relu and h._want_fwd_absmax():
        torch.cuda.synchronize()
        assert h.absmax_value(y._egz_absmax).item() == y.max().item()


def test_cabi_argument_errors_are_loud():
    """Error behaviour of the boundary: bad arguments never launch -- the C-ABI returns a non-zero code with a message
    (egz_last_error) and the Python layer raises RuntimeError, as the reference's torch ops would (empty batch, wrong
    layout, host tensors, a geometry an entry point does not cover, an undersized workspace)."""
    h = H()
    w = rnd(64, 64, 3, 3, seed=1).to(DEV)
    x = nhwc(rnd(2, 64, 16, 16, seed=2))
    wp = h.packed_weight(w, "fwd", 0)
    with pytest.raises(RuntimeError):                                  # empty batch
        h.conv3x3_fwd(torch.empty((0, 16, 16, 64), device=DEV), wp, None, 64)
    with pytest.raises(RuntimeError):                                  # non-contiguous operand
        h.conv3x3_fwd(x.transpose(1, 2), wp, None, 64)
    with pytest.raises(RuntimeError):                                  # host tensor
        h.conv3x3_fwd(x.cpu(), wp, None, 64)
    with pytest.raises(RuntimeError):                                  # fp64 operand
        h.conv3x3_fwd(x.double(), wp, None, 64)
    # a geometry the streamed kernel does not cover is refused by the entry point itself, not silently mis-run
    assert h.LIB.egz_conv3x3_streamed_ok(2, 16, 16, 30, 64, 0) == 0
    y = torch.empty((2, 16, 16, 64), device=DEV)
    rc = h.LIB.egz_conv3x3_fwd_streamed(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), None, 2, 16, 16, 30, 64, 0, 1, 0,
                                        None, None, None, None, None, h._stream())
    assert rc != 0
    with pytest.raises(RuntimeError, match="egz_conv3x3_fwd_streamed"):
        h.check(rc, "egz_conv3x3_fwd_streamed")
    # the BatchNorm-sums epilogue exists for the narrow geometry and for 64- / 128-column tiles: not for 96 columns
    rc = h.LIB.egz_conv3x3_fwd_streamed(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), y.data_ptr(), 2, 16, 16, 64, 96, 5, 1,
                                        0, None, y.data_ptr(), None, y.data_ptr(), None, h._stream())
    with pytest.raises(RuntimeError, match="narrow geometry"):
        h.check(rc, "egz_conv3x3_fwd_streamed")
    # split-K needs its workspace
    xs = nhwc(rnd(1, 256, 14, 14, seed=3))
    ws_ = rnd(512, 256, 3, 3, seed=4).to(DEV)
    wq, st = h.conv_weight(ws_, "fwd", 1, xs, 512)
    ns = h.LIB.egz_conv3x3_streamed_splits(1, 14, 14, 256, 512)
    assert st and ns >= 2
    yo = torch.empty((1, 14, 14, 512), device=DEV)
    small = torch.empty(16, device=DEV)
    rc = h.LIB.egz_conv3x3_fwd_streamed_splitk(xs.data_ptr(), wq.data_ptr(), None, yo.data_ptr(), None, 1, 14, 14, 256, 512, 0, 1,
                                               None, small.data_ptr(), 64, ns, None, h._stream())
    assert rc != 0
    with pytest.raises(RuntimeError, match="workspace"):
        h.check(rc, "egz_conv3x3_fwd_streamed_splitk")


def test_large_operand_routes_to_the_64bit_addressed_kernels(monkeypatch):
    """Operands of 4 GiB or more cannot use the split-half kernels' 32-bit buffer offsets: conv_dtype sends them to the
    exact-f32 kernels (64-bit addressing).  The threshold is lowered here so that an ordinary tensor takes that route, and
    the result is checked against the default route."""
    h = H()
    if h.PRECISION != "split":
        pytest.skip("routing rule of the split-half mode (default)")
    x = nhwc(rnd(2, 64, 32, 32, seed=5))
    assert h.conv_dtype("fwd", 128, 64, x) == h.F16X3
    monkeypatch.setattr(h, "_SPLIT_MAX_BYTES", 1024)
    assert h.conv_dtype("fwd", 128, 64, x) == h.F32 and h.conv_dtype("dgrad", 64, 128, x) == h.F32


def test_pack_frag_batch_matches_per_layer(monkeypatch):
    """egz_pack_w3x3_frag_batch (every stale fragment-ordered packing of an optimizer step in one launch, FusedAdam.step ->
    hipops.refresh_packings) writes the same bytes as the per-layer egz_pack_w3x3_split_frag launches: all four kinds, both split
    dtypes, full and padded (C or K < 32, unaligned base pointer) tensors; packings nobody used since their last rebuild stay
    stale; ``force`` rebuilds fresh ones too."""
    h = H()
    shapes = [(64, 64), (128, 64), (64, 128), (512, 256), (32, 32), (32, 8), (8, 32), (20, 64)]
    flat = torch.empty(sum(K * C * 9 for C, K in shapes) + 3, device=DEV)
    ws, off = [], 3                                   # offset 3 floats: base pointers that are not 16-byte aligned
    for i, (C, K) in enumerate(shapes):
        w = flat[off:off + K * C * 9].view(K, C, 3, 3)
        w.copy_(rnd(K, C, 3, 3, seed=40 + i, scale=0.05))
        off += K * C * 9
        ws.append(w)
    kinds = [("fwd_frag", 1), ("dgrad_frag", 1), ("dgrad_frag", 2), ("ups_fwd_frag", 1), ("ups_dgrad_frag", 1), ("ups_dgrad_frag", 2)]

    def pack_all():
        return {(i, k, dt): h.packed_weight(w, k, dt) for i, w in enumerate(ws) for k, dt in kinds}

    monkeypatch.setattr(h, "BATCH_REPACK", True)
    first = pack_all()                                # lazy per-layer launches (nothing cached yet)
    for w in ws:
        w.mul_(1.7)                                   # version bump: every packing is stale now
    assert h.refresh_packings(ws) == len(first)       # ONE launch
    batch = {k: v.clone() for k, v in pack_all().items()}          # cache hits: the buffers the batch launch wrote
    assert all(batch[k].data_ptr() != first[k].data_ptr() for k in first)
    assert h.refresh_packings(ws) == 0                # fresh and not used since -> pack_all marked them used, but they are fresh
    for w in ws:
        w.add_(0.0)                                   # stale again, values unchanged
    monkeypatch.setattr(h, "BATCH_REPACK", False)
    assert h.refresh_packings(ws) == 0
    lazy = pack_all()                                 # per-layer rebuilds of the same values
    for k in lazy:
        assert torch.equal(lazy[k].view(torch.int32), batch[k].view(torch.int32)), k
    monkeypatch.setattr(h, "BATCH_REPACK", True)
    want = {k: v.clone() for k, v in lazy.items()}
    for v in lazy.values():
        v.zero_()                                     # fresh by their tags, wrong by their bytes
    assert h.refresh_packings(ws) == 0
    assert h.refresh_packings(ws, force=True) == len(first)        # force rebuilds regardless of tags
    for k in lazy:
        assert torch.equal(lazy[k].view(torch.int32), want[k].view(torch.int32)), k


@pytest.mark.parametrize("B,Hh,Ww,C,K,pool", [
    (2, 32, 32, 64, 64, False),       # 256 x 64 tile, 16 x 16 patches; weight gradient on 4 x 8 patches? (W < 112: 1 x 32)
    (1, 112, 112, 64, 128, True),     # pooled producer -> 56 x 56 consumer
    (2, 28, 28, 128, 256, False),     # raster-run geometry, masked 8-wide weight-gradient patches
    (3, 14, 14, 256, 128, False),     # 14 x 14: raster run, 2 x 16 masked patches
    (1, 224, 224, 64, 64, False),     # the 224-wide layers: 4 x 8 patches
    (2, 56, 56, 128, 64, True),       # pooled to 28 x 28, 64-column consumer
])
def test_presplit_activation_chain(B, Hh, Ww, C, K, pool, monkeypatch):
    """Pre-split activations (hipops.PRESPLIT, round 5): conv (statistics epilogue + per-channel max / min) -> egz_bn_finalize_bound
    -> egz_bn_relu_pool_fwd_presplit -> consumer conv forward (mode | 0x100) and weight gradient (flags | 0x8000).
    (a) the bound is the EXACT maximum of the block output (== the abs-max the fp32 form measures in its own pass);
    (b) the stored pairs are exactly the pairs the consumers form while staging the fp32 tensor, so the consumer's forward
        output, its BatchNorm partial sums and its weight gradient are BIT-IDENTICAL to the fp32-activation path."""
    h = H()
    monkeypatch.setattr(h, "SPLITK", False)              # (the small test shapes would otherwise take split-K launches)
    x0 = rnd(B, 64, Hh, Ww, seed=301)
    w0 = rnd(C, 64, 3, 3, seed=302, scale=(2.0 / (9 * 64)) ** 0.5)
    b0 = rnd(C, seed=303, scale=0.1)
    gam, bet = 1.0 + 0.3 * rnd(C, seed=304), 0.2 * rnd(C, seed=305)
    x0d, w0d = nhwc(x0), w0.to(DEV)
    wp0, st0 = h.conv_weight(w0d, "fwd", h.F16X3, x0d, C)
    assert st0
    y, stat = h.conv3x3_fwd(x0d, wp0, b0.to(DEV), C, epi=h.EPI_BIAS_STATS, dtype=h.F16X3, streamed=True, want_bound=True)
    mm = y._egz_mm
    # the per-channel max / min images against the tensor itself
    yc = y.reshape(-1, C)
    nsl = max(512 // C, 1)                                # slot sets of 2 C uints; unused ones stay zero (= lowest)
    u = mm[:nsl * 2 * C].view(torch.int32).cpu().numpy().astype(np.uint32).reshape(nsl, 2 * C).max(0)
    dec = np.where(u & 0x80000000, u ^ np.uint32(0x80000000), ~u).astype(np.uint32).view(np.float32)
    assert np.array_equal(dec[:C], yc.max(0).values.cpu().numpy())
    assert np.array_equal(-dec[C:2 * C], yc.min(0).values.cpu().numpy())
    n = float(B * Hh * Ww)
    coef, am = h.bn_finalize(stat, n, gam.to(DEV), bet.to(DEV), None, None, 0.1, 1e-5, mm=mm)
    coef_ref = h.bn_finalize(stat, n, gam.to(DEV), bet.to(DEV), None, None, 0.1, 1e-5)
    assert torch.equal(coef, coef_ref)
    out_ref = h.bn_relu_pool_fwd(y, coef, pool)
    assert float(h.absmax_value(am)) == float(h.absmax_value(out_ref._egz_absmax)) == float(out_ref.max())
    out_pre = h.bn_relu_pool_fwd(y, coef, pool, presplit_am=am)
    assert out_pre._egz_presplit and out_pre.shape == out_ref.shape
    Ho, Wo = out_ref.shape[1], out_ref.shape[2]
    assert h.presplit_ok(B, Ho, Wo, C, K)
    # consumer: forward + statistics
    w1 = rnd(K, C, 3, 3, seed=306, scale=(2.0 / (9 * C)) ** 0.5).to(DEV)
    b1 = rnd(K, seed=307, scale=0.1).to(DEV)
    wp1, st1 = h.conv_weight(w1, "fwd", h.F16X3, out_ref, K)
    assert st1
    y_ref, s_ref = h.conv3x3_fwd(out_ref, wp1, b1, K, epi=h.EPI_BIAS_STATS, dtype=h.F16X3, streamed=True)
    y_pre, s_pre = h.conv3x3_fwd(out_pre, wp1, b1, K, epi=h.EPI_BIAS_STATS, dtype=h.F16X3, streamed=True, pre_in=True)
    assert torch.equal(y_ref, y_pre) and torch.equal(s_ref, s_pre)
    # consumer: weight gradient
    dy = nhwc(rnd(B, K, Ho, Wo, seed=308, scale=1e-3))
    dw_ref = h.conv3x3_wgrad(out_ref, dy, precision="split_f16")
    dw_pre = h.conv3x3_wgrad(out_pre, dy, precision="split_f16", x_pre=True)
    assert torch.equal(dw_ref, dw_pre)
    # two products per MAC (the opt-in, EGAZE_BWD_PRODUCTS=2): x enters hi-only, rounded to nearest from the fp32 value in one launch
    # and from hi + lo of the stored pair in the other -- a double rounding apart on rare ties: close, not bit-identical
    monkeypatch.setattr(h, "BWD_PRODUCTS", 2)
    assert rel(h.conv3x3_wgrad(out_pre, dy, precision="split_f16", x_pre=True), h.conv3x3_wgrad(out_ref, dy, precision="split_f16")) < 1e-4       # observed 2.5e-5
    monkeypatch.setattr(h, "BWD_PRODUCTS", 3)
    # a consumer that cannot take the pairs refuses them
    with pytest.raises(RuntimeError):
        h.conv3x3_fwd(out_pre, wp1, b1, K, epi=h.EPI_BIAS_RELU, dtype=h.F16X3, streamed=True, pre_in=True)


HEADLINE_SHAPES = [        # (Cin, Cout, H (output), upsampled): the 12 distinct conv geometries of the SP step
    (64, 64, 224, False), (64, 128, 112, False), (128, 128, 112, False), (128, 256, 56, False), (256, 256, 56, False),
    (256, 512, 28, False), (512, 512, 28, False), (512, 512, 14, False), (512, 256, 56, True), (256, 128, 112, True),
    (128, 64, 224, True), (64, 64, 224, False),
]


@pytest.mark.parametrize("C,K,Hh,ups", HEADLINE_SHAPES[:11])
def test_conv_ops_elementwise_at_the_headline_geometry(C, K, Hh, ups, monkeypatch):
    """VERDICT r4 (parity soft spot a): every element-wise gradient check of the whole model runs at 32 x 32 with split-K pinned;
    at the headline geometry (batch 32, 224 x 224) the whole-model comparison is limited by ReLU / max-pool subgradient flips
    (tests/report_headline_grads.py).  Here each convolution of the step is checked ELEMENT-WISE in exactly the launch geometry
    bench.py times -- batch 32, the real image sizes (3136 / 6272 / 1568 / ... tiles per launch, the weight gradient's real
    split-K depth, default SPLITK decision) -- against torch-CPU fp32 on the same operands: forward, data gradient and weight
    gradient, every entry within 2e-5 (5e-5 for the 1.6 M-pixel weight-gradient reductions) of max |ref| with three products per
    MAC; the two-product backward arithmetic (hipops.BWD_PRODUCTS = 2, opt-in) on the same launches: every entry within 2e-3,
    relative L2 error below 1e-3.  A mis-indexed tile moves 1 / 3136 of the entries by O(1)."""
    h = H()
    B = 32
    keep = torch.get_num_threads()
    torch.set_num_threads(min(32, __import__("os").cpu_count() or 1))
    try:
        hin = Hh // 2 if ups else Hh
        x = rnd(B, C, hin, hin, seed=401).clamp_(min=0)                  # post-ReLU-like operand
        w = rnd(K, C, 3, 3, seed=402, scale=(2.0 / (9 * C)) ** 0.5)
        b = rnd(K, seed=403, scale=0.1)
        dy = rnd(B, K, Hh, Hh, seed=404, scale=1e-3)
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups else xr
        ref = F.conv2d(xin, wr, b, padding=1)
        ref.backward(dy)
        xd, dyd, wd = nhwc(x), nhwc(dy), w.to(DEV)
        dt = h.conv_dtype("fwd", K, C, xd)
        wp, st = h.conv_weight(wd, "ups_fwd" if ups else "fwd", dt, xd, K)
        assert dt == h.F16X3 and st
        y, _ = h.conv3x3_fwd(xd, wp, b.to(DEV), K, ups="phase" if ups else False, epi=h.EPI_BIAS_RELU if ups else h.EPI_BIAS,
                             dtype=dt, streamed=st)
        want = F.relu(ref.detach()) if ups else ref.detach()
        e_f = rel(nchw(y), want)
        ddt = h.conv_dtype("dgrad", C, K, dyd)
        wq, sq = h.conv_weight(wd, "ups_dgrad" if ups else "dgrad", ddt, dyd, C)
        err = {}
        for products in (3, 2):           # backward arithmetic: three MFMA products per MAC (fp32 class, the default), two (opt-in)
            monkeypatch.setattr(h, "BWD_PRODUCTS", products)
            dx = h.conv3x3_ups_dgrad(dyd, wq, C, dtype=ddt, streamed=sq) if ups else h.conv3x3_dgrad(dyd, wq, C, dtype=ddt, streamed=sq)
            dw = h.conv3x3_wgrad(xd, dyd, ups=ups)
            err[products] = (rel(nchw(dx), xr.grad), rel(dw.cpu(), wr.grad),
                             float((nchw(dx).double() - xr.grad.double()).norm() / xr.grad.double().norm()),
                             float((dw.cpu().double() - wr.grad.double()).norm() / wr.grad.double().norm()))
        (e_d, e_w, _, _), (p_d, p_w, l_d, l_w) = err[3], err[2]
        print(f"B=32 {C}->{K} @{Hh}{'u' if ups else ''}: fwd {e_f:.1e}  dgrad {e_d:.1e}  wgrad {e_w:.1e}  | two products: "
              f"dgrad {p_d:.1e} (L2 {l_d:.1e})  wgrad {p_w:.1e} (L2 {l_w:.1e})")
        assert e_f < 2e-5 and e_d < 2e-5 and e_w < 5e-5, (e_f, e_d, e_w)
        # two products: one operand enters with 11 significant bits, rounded to nearest (dy in the data gradient, x in the weight
        # gradient): every entry within 2e-3 of max |ref|, relative L2 error below 1e-3 (a mis-indexed
        # tile would still show as O(1) entries)
        assert p_d < 2e-3 and p_w < 2e-3 and l_d < 1e-3 and l_w < 1e-3, (p_d, p_w, l_d, l_w)
    finally:
        torch.set_num_threads(keep)


@pytest.mark.parametrize("B,Hh,Ww,C,K,pool", [
    (2, 32, 32, 64, 64, False), (1, 112, 112, 64, 128, True), (2, 28, 28, 128, 256, False), (3, 14, 14, 256, 128, False),
    (1, 224, 224, 64, 64, False), (2, 56, 56, 128, 64, True),
])
def test_presplit_gradient_chain(B, Hh, Ww, C, K, pool, monkeypatch):
    """Pre-split GRADIENTS (hipops.PRESPLIT_GRAD): the BatchNorm backward of a C -> K block writes dy as f16 pairs scaled by a
    bound of max |dy| derived in its finalize step (from max |dout|, the per-channel max / min of y and the two sums).
    (a) the bound holds and is tight: max |dy| <= bound <= 8 max |dy|;  (b) dgamma / dbeta are untouched (torch.equal);
    (c) the consumers -- data gradient (plain and with the BatchNorm-sums epilogue) and weight gradient -- agree with the
    fp32-gradient launches to 2e-6 of max |ref| and with fp64 like them."""
    h = H()
    monkeypatch.setattr(h, "SPLITK", False)
    x0 = rnd(B, C, Hh, Ww, seed=501).clamp_(min=0)
    w = rnd(K, C, 3, 3, seed=502, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(K, seed=503, scale=0.1)
    gam, bet = 1.0 + 0.3 * rnd(K, seed=504), 0.2 * rnd(K, seed=505)
    xd, wd = nhwc(x0), w.to(DEV)
    wp, st = h.conv_weight(wd, "fwd", h.F16X3, xd, K)
    y, stat = h.conv3x3_fwd(xd, wp, b.to(DEV), K, epi=h.EPI_BIAS_STATS, dtype=h.F16X3, streamed=st, want_bound=True)
    mm = y._egz_mm
    coef = h.bn_finalize(stat, float(B * Hh * Ww), gam.to(DEV), bet.to(DEV), None, None, 0.1, 1e-5)
    Ho, Wo = (Hh // 2, Ww // 2) if pool else (Hh, Ww)
    dout = nhwc(rnd(B, K, Ho, Wo, seed=506, scale=3e-4))
    dout_am = h.absmax_of(dout)
    assert h.presplit_grad_ok(B, Hh, Ww, C, K)
    dy_ref, dg_ref, db_ref = h.bn_relu_pool_bwd(y, dout, coef, pool)
    dy_pre, dg_pre, db_pre = h.bn_relu_pool_bwd(y, dout, coef, pool, presplit=(dout_am, mm))
    assert torch.equal(dg_ref, dg_pre) and torch.equal(db_ref, db_pre)
    true_max, bound = float(dy_ref.abs().max()), float(h.absmax_value(dy_pre._egz_absmax))
    print(f"max |dy| {true_max:.3e}, bound {bound:.3e} ({bound / true_max:.2f}x)")
    assert true_max <= bound <= 8 * true_max
    wq, sq = h.conv_weight(wd, "dgrad", h.F16X3, dy_ref, C)
    dx_ref = h.conv3x3_dgrad(dy_ref, wq, C, dtype=h.F16X3, streamed=sq)
    dx_pre = h.conv3x3_dgrad(dy_pre, wq, C, dtype=h.F16X3, streamed=sq, pre_in=True)
    assert rel(dx_pre, dx_ref) < 2e-6
    assert float(h.absmax_value(dx_pre._egz_absmax)) == float(dx_pre.abs().max())        # the data gradient's own abs-max
    dw_ref = h.conv3x3_wgrad(xd, dy_ref, precision="split_f16")
    dw_pre = h.conv3x3_wgrad(xd, dy_pre, precision="split_f16", dy_pre=True)
    assert rel(dw_pre, dw_ref) < 2e-6
    # against fp64 (the bound-scaled pairs are as accurate as the abs-max-scaled ones)
    dyc = nchw(dy_ref).double()
    wref = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    xin = x0.double().requires_grad_(True)
    F.conv2d(xin, wref, None, padding=1).backward(dyc)
    dxt = torch.nn.grad.conv2d_input(x0.shape, w.double(), dyc, padding=1)
    assert rel(nchw(dx_pre), dxt) < 2e-6 and rel(dw_pre.cpu(), wref.grad) < 2e-6
    if not pool and C % 64 == 0:
        # the BatchNorm-sums epilogue over pairs: dx and the two sums of the block BELOW (its y / coefficients: reuse x0's role)
        ybelow = nhwc(rnd(B, C, Hh, Ww, seed=507))
        cbelow = torch.stack([0.1 * rnd(C, seed=508), 1.0 + 0.1 * rnd(C, seed=509).abs(), 1.0 + 0.2 * rnd(C, seed=510), 0.1 * rnd(C, seed=511)]).to(DEV).contiguous()
        if h.bnsums_ok(B, Hh, Ww, C, K, h.F16X3):
            d1, s1 = h.conv3x3_dgrad_bnsums(dy_ref, wq, C, h.F16X3, ybelow, cbelow)
            d2, s2 = h.conv3x3_dgrad_bnsums(dy_pre, wq, C, h.F16X3, ybelow, cbelow, pre_in=True)
            assert rel(d2, d1) < 2e-6 and rel(s2.sum(0), s1.sum(0)) < 1e-5
    # (d) the same consumers on two products per MAC (hipops.BWD_PRODUCTS = 2, opt-in): the pairs serve as
    # the hi-only operand of the data gradient (rounded to nearest from hi + lo) and as the 22-bit operand of the weight gradient; both stay in the two-product error class against the three-product results
    monkeypatch.setattr(h, "BWD_PRODUCTS", 2)
    for name, got, want in (("dgrad", h.conv3x3_dgrad(dy_pre, wq, C, dtype=h.F16X3, streamed=sq, pre_in=True), dx_ref),
                            ("wgrad", h.conv3x3_wgrad(xd, dy_pre, precision="split_f16", dy_pre=True), dw_ref),
                            ("wgrad, in-kernel split", h.conv3x3_wgrad(xd, dy_ref, precision="split_f16"), dw_ref)):
        l2 = float((got.double() - want.double()).norm() / want.double().norm())
        assert 1e-6 < l2 < 1e-3 and rel(got, want) < 2e-3, (name, l2, rel(got, want))


@pytest.mark.parametrize("B,Hh,Ww,C,K,ups", [(2, 28, 28, 64, 128, False), (1, 56, 56, 128, 64, True), (3, 14, 14, 256, 256, False),
                                             (2, 32, 32, 64, 64, False)])
def test_backward_two_products(B, Hh, Ww, C, K, ups, monkeypatch):
    """hipops.BWD_PRODUCTS = 2 (opt-in; 3 is the default): the backward convolutions issue a_hi b_hi + a_lo b_hi -- two MFMA products per MAC,
    the halo operand (dy / x) with its f16 hi half only, rounded to nearest (csrc/egz_common.h, egz_f16p2).  Against torch fp64 on the same operands:
    (a) the FORWARD launch does not know the knob (bit-identical output);  (b) data and weight gradient stay within 2e-3 of
    max |ref| per entry and 1e-3 in relative L2 -- and are NOT the three-product results (the knob reaches the launches);
    (c) three products: the fp32 class (2e-5).  Plain and upsample-fused geometries, 64- and 128-column tiles."""
    h = H()
    monkeypatch.setattr(h, "SPLITK", False)       # (split-K launches of few-tile geometries stay three-product)
    hin, win = (Hh // 2, Ww // 2) if ups else (Hh, Ww)
    x = rnd(B, C, hin, win, seed=31).clamp_(min=0)
    w = rnd(K, C, 3, 3, seed=32, scale=(2.0 / (9 * C)) ** 0.5)
    dy = rnd(B, K, Hh, Ww, seed=33, scale=1e-3)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups else xr
    F.conv2d(xin, wr, None, padding=1).backward(dy.double())
    xd, dyd, wd = nhwc(x), nhwc(dy), w.to(DEV)
    dt = h.conv_dtype("fwd", K, C, xd)
    wp, st = h.conv_weight(wd, "ups_fwd" if ups else "fwd", dt, xd, K)
    ddt = h.conv_dtype("dgrad", C, K, dyd)
    wq, sq = h.conv_weight(wd, "ups_dgrad" if ups else "dgrad", ddt, dyd, C)
    assert dt == h.F16X3 and ddt == h.F16X3 and st and sq
    got = {}
    for products in (3, 2):
        monkeypatch.setattr(h, "BWD_PRODUCTS", products)
        y, _ = h.conv3x3_fwd(xd, wp, None, K, ups="phase" if ups else False, epi=h.EPI_BIAS_RELU if ups else h.EPI_BIAS, dtype=dt, streamed=st)
        dx = h.conv3x3_ups_dgrad(dyd, wq, C, dtype=ddt, streamed=sq) if ups else h.conv3x3_dgrad(dyd, wq, C, dtype=ddt, streamed=sq)
        dw = h.conv3x3_wgrad(xd, dyd, ups=ups)
        got[products] = (y, dx, dw)
    assert torch.equal(got[2][0], got[3][0])
    for products, tol_max, tol_l2 in ((3, 2e-5, 2e-6), (2, 2e-3, 1e-3)):
        _, dx, dw = got[products]
        for name, a, ref in (("dgrad", nchw(dx), xr.grad), ("wgrad", dw.cpu(), wr.grad)):
            l2 = float((a.double() - ref).norm() / ref.norm())
            assert rel(a, ref) < tol_max and l2 < tol_l2, (products, name, rel(a, ref), l2)
    assert not torch.equal(got[2][1], got[3][1]) and not torch.equal(got[2][2], got[3][2])
