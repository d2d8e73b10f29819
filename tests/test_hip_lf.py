"""GPU parity of the LF path (late_fusion + floss through the C-ABI) against the reference's golden vectors
(tests/golden/late_fusion.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import egaze_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build():
    from egaze_amd.models.late_fusion import late_fusion
    net = late_fusion()
    net.load_state_dict(synth.synth_state_dict(O.lf_shapes(), seed=3, head_gain=0.5))
    return net.to(DEV)


@pytest.mark.parametrize("C,K", [(32, 32), (32, 8), (8, 32)])
def test_narrow_conv_ops(C, K):
    """The 32-wide igemm tile, padded channel counts and the 32x32 wgrad tile used by late_fusion."""
    import egaze_amd.hipops as h
    g = torch.Generator().manual_seed(C * 100 + K)
    x = torch.randn(2, C, 13, 10, generator=g).requires_grad_(True)
    w = (torch.randn(K, C, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(K, generator=g) * 0.1
    ref = F.conv2d(x, w, b, padding=1)
    dy = torch.randn(2, K, 13, 10, generator=g)
    ref.backward(dy)
    wd = w.detach().to(DEV)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    y, stat = h.conv3x3_fwd(xd, h.packed_weight(wd, "fwd"), b.to(DEV), K, epi=h.EPI_BIAS_STATS)
    assert rel(y.permute(0, 3, 1, 2).cpu().numpy(), ref.detach().numpy()) < 2e-5
    assert rel(stat.sum(0)[0].cpu().numpy(), ref.detach().double().sum(dim=(0, 2, 3)).numpy()) < 1e-5
    dx = h.conv3x3_dgrad(dyd, h.packed_weight(wd, "dgrad"), C)
    assert rel(dx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.numpy()) < 2e-5
    dw = h.conv3x3_wgrad(xd, dyd)
    assert rel(dw.cpu().numpy(), w.grad.numpy()) < 2e-5


@pytest.mark.parametrize("tag,size", [("s32", 32), ("s224", 224)])
def test_late_fusion_golden(tag, size):
    from egaze_amd.floss import floss
    gold = np.load(os.path.join(GOLDEN, "late_fusion.npz"))
    net = build()
    im, feat, gt = synth.synth_lf_batch(2, size, seed=7)
    imd, featd, gtd = im.to(DEV), feat.to(DEV), gt.to(DEV)
    net.eval()
    with torch.no_grad():
        ev = net(featd, imd)
        sw = net(imd, featd)
    assert rel(ev.cpu().numpy(), gold[f"{tag}_eval_out"]) < 2e-5
    assert abs(sw.double().sum().item() - float(gold[f"{tag}_eval_out_swapped_sum"])) < 1e-4 * abs(float(gold[f"{tag}_eval_out_swapped_sum"]))
    net.train()
    out = net(featd, imd)                                  # LF.py:90 argument order
    loss = floss()(out, gtd)
    loss.backward()
    assert rel(out.detach().cpu().numpy(), gold[f"{tag}_train_out"]) < 5e-5
    assert abs(loss.item() - float(gold[f"{tag}_loss"])) < 1e-4 * abs(float(gold[f"{tag}_loss"]))
    gmax = max(np.abs(gold[f"{tag}_grad/{k}"]).max() for k, _ in net.named_parameters())
    for k, p in net.named_parameters():
        want = gold[f"{tag}_grad/{k}"]
        if np.abs(want).max() < 1e-5 * gmax:               # conv biases in front of train-mode BN: exactly 0 here
            assert p.grad.abs().max().item() < 1e-4 * gmax, k
            continue
        a, b = p.grad.cpu().numpy().ravel().astype(np.float64), want.ravel().astype(np.float64)
        l2 = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert l2 < 2e-2 and rel(a, b) < 0.1, (k, l2, rel(a, b))
    for k, v in net.state_dict().items():
        if "running_" in k:
            assert rel(v.cpu().numpy(), gold[f"{tag}_after/{k}"]) < 1e-4, k


def test_late_fusion_grads_vs_fp64():
    from egaze_amd.floss import floss
    net = build()
    sd = synth.synth_state_dict(O.lf_shapes(), seed=3, head_gain=0.5)
    im, feat, gt = synth.synth_lf_batch(3, 48, seed=11)
    net.train()
    out = net(feat.to(DEV), im.to(DEV))
    floss()(out, gt.to(DEV)).backward()

    def run(dtype):
        work = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        keys = O.trainable_keys(work)
        for k in keys:
            work[k].requires_grad_(True)
        o = O.late_fusion_forward(work, feat.to(dtype), im.to(dtype), training=True)
        O.floss_forward(o, gt.to(dtype)).backward()
        return o.detach(), {k: work[k].grad for k in keys}
    o32, g32 = run(torch.float32)
    o64, g64 = run(torch.float64)
    assert rel(out.detach().cpu().numpy(), o64.numpy()) < max(5 * rel(o32.numpy(), o64.numpy()), 2e-6)
    eh, ec = [], []
    for k, p in net.named_parameters():
        if g64[k].abs().max().item() < 1e-9:
            continue
        eh.append(rel(p.grad.cpu().numpy(), g64[k].numpy()))
        ec.append(rel(g32[k].numpy(), g64[k].numpy()))
    print("LF HIP vs fp64: median %.2e max %.2e | CPU fp32: median %.2e max %.2e" %
          (np.median(eh), max(eh), np.median(ec), max(ec)))
    assert np.median(eh) < max(20 * np.median(ec), 2e-4) and max(eh) < 5e-2


def test_bn_sums_folded_into_narrow_dgrad():
    """The narrow data-gradient kernel accumulates the BatchNorm-backward sums of the layer below (conv3x3_dgrad_bnsums):
    per-channel sums against an fp64 torch evaluation, and the whole LF backward against the path with the separate
    reduce pass (same kernels otherwise; only the summation order of those two sums differs)."""
    from egaze_amd import hipops as H
    from egaze_amd.floss import floss
    if H.PRECISION != "split":
        pytest.skip("the folded BatchNorm sums live in the split-half narrow kernel (default mode)")
    torch.manual_seed(5)
    B, S, C, K = 2, 32, 32, 32
    dy = torch.randn(B, S, S, K, device=DEV) * 1e-3
    w = torch.randn(K, C, 3, 3, device=DEV) * 0.1
    bn_y = torch.randn(B, S, S, C, device=DEV)
    coef = torch.stack([bn_y.mean((0, 1, 2)), 1.0 / bn_y.std((0, 1, 2)), torch.rand(C, device=DEV) + 0.5,
                        torch.randn(C, device=DEV) * 0.3]).contiguous()
    wp, st = H.conv_weight(w, "dgrad", H.F16X3, dy, C)
    assert st
    dx, sums = H.conv3x3_dgrad_bnsums(dy, wp, C, H.F16X3, bn_y, coef)
    ref = H.conv3x3_dgrad(dy, wp, C, dtype=H.F16X3, streamed=st)
    assert torch.equal(dx, ref)
    d64, y64, c64 = dx.double(), bn_y.double(), coef.double()
    dz = torch.where(y64 * c64[2] + c64[3] > 0, d64, torch.zeros_like(d64))
    s = sums.sum(0)
    assert rel(s[0].cpu().numpy(), dz.sum((0, 1, 2)).cpu().numpy()) < 1e-6
    assert rel(s[1].cpu().numpy(), (dz * (y64 - c64[0]) * c64[1]).sum((0, 1, 2)).cpu().numpy()) < 1e-5

    im, feat, gt = synth.synth_lf_batch(3, 48, seed=11)
    grads = {}
    for fuse in (True, False):
        H.BNSUMS_FUSE = fuse
        try:
            net = build()
            net.train()
            before = dict(H.BNSUMS_STATS)
            floss()(net(feat.to(DEV), im.to(DEV)), gt.to(DEV)).backward()
            torch.cuda.synchronize()
            made = H.BNSUMS_STATS["produced"] - before["produced"], H.BNSUMS_STATS["consumed"] - before["consumed"]
            assert made == ((2, 2) if fuse else (0, 0)), made      # the 32 -> 32 and 32 -> 8 blocks feed the two below them
            grads[fuse] = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()}
        finally:
            H.BNSUMS_FUSE = True
    worst = max(rel(grads[True][k], grads[False][k]) for k in grads[True] if np.abs(grads[False][k]).max() > 1e-9)
    print("LF grads, fused BN sums vs separate reduce pass: max rel %.2e" % worst)
    assert worst < 2e-5


def test_deferred_batchnorm_matches_materialised():
    """Training step of the late-fusion stack with the [BN -> ReLU] of the two 32-channel blocks applied by the NEXT block's
    conv / weight-gradient kernels while they stage the pre-BN tensor (hipops.BN_DEFER) against the same step with the
    normalised tensors materialised: identical arithmetic per element (same fma, same exact abs-max, hence the same f16 split),
    so outputs, running statistics and gradients must agree to fp32 round-off of the reductions."""
    from egaze_amd import hipops as H
    from egaze_amd.floss import floss
    if H.PRECISION != "split":
        pytest.skip("deferred BatchNorm lives in the split-half narrow kernels (default mode)")
    im, feat, gt = synth.synth_lf_batch(3, 48, seed=11)
    res = {}
    for defer in (True, False):
        H.BN_DEFER = defer
        try:
            net = build()
            net.train()
            before = H.BN_DEFER_STATS["deferred"]
            out = net(feat.to(DEV), im.to(DEV))
            floss()(out, gt.to(DEV)).backward()
            torch.cuda.synchronize()
            assert H.BN_DEFER_STATS["deferred"] - before == (2 if defer else 0)      # blocks 1 and 2; block 3 feeds the 1x1 head
            res[defer] = (out.detach().cpu().numpy().copy(),
                          {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()},
                          {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items() if "running" in k})
        finally:
            H.BN_DEFER = True
    o1, g1, r1 = res[True]
    o0, g0, r0 = res[False]
    print("deferred BN: out max abs diff %.2e" % np.abs(o1 - o0).max())
    assert rel(o1, o0) < 1e-6
    for k in r0:
        assert rel(r1[k], r0[k]) < 1e-6, k
    worst = max(rel(g1[k], g0[k]) for k in g0 if np.abs(g0[k]).max() > 1e-9)
    print("deferred BN: grads max rel %.2e" % worst)
    assert worst < 2e-5


@pytest.mark.parametrize("B,C,Hh,Ww,K", [(2, 2, 48, 48, 32), (3, 2, 13, 20, 32), (1, 3, 40, 72, 32), (2, 1, 16, 16, 32), (2, 2, 9, 13, 32),
                                         (1, 2, 224, 224, 32), (2, 3, 48, 48, 64), (1, 3, 13, 21, 64), (1, 3, 224, 224, 64), (2, 2, 16, 20, 64)])
def test_first_block_backward_in_one_pass(B, C, Hh, Ww, K):
    """bn_bwd_first_wgrad (BatchNorm backward + weight gradient of the first late-fusion conv without storing the gradient
    w.r.t. the conv output) against fp64 autograd of Conv2d(C -> 32) -> BatchNorm2d(train) -> ReLU; ragged sizes (rows shorter
    than the 32-pixel step, a partial last block)."""
    from egaze_amd import hipops as H
    g = torch.Generator().manual_seed(B * 1000 + Hh)
    x = torch.randn(B, C, Hh, Ww, generator=g)
    w = (torch.randn(K, C, 3, 3, generator=g) * 0.3).double().requires_grad_(True)
    gamma = (torch.rand(K, generator=g) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(K, generator=g) * 0.2).double().requires_grad_(True)
    dout = torch.randn(B, K, Hh, Ww, generator=g)
    y = F.conv2d(x.double(), w, None, padding=1)
    out = F.relu(F.batch_norm(y, None, None, gamma, beta, training=True, eps=1e-5))
    out.backward(dout.double())
    yd = y.detach().float().permute(0, 2, 3, 1).contiguous().to(DEV)
    mean = y.detach().mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.detach().var((0, 2, 3), unbiased=False) + 1e-5)
    coef = torch.stack([mean, invstd, gamma.detach() * invstd, beta.detach() - mean * gamma.detach() * invstd]).float().to(DEV)
    dd = dout.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw, dg, db = H.bn_bwd_first_wgrad(yd, dd, coef, x.to(DEV))
    assert rel(dw.cpu().numpy(), w.grad.numpy()) < 2e-5
    assert rel(dg.cpu().numpy(), gamma.grad.numpy()) < 2e-5 and rel(db.cpu().numpy(), beta.grad.numpy()) < 2e-5
    # same numbers as the two-pass route (apply pass, then the generic first-layer weight gradient) up to fp32 summation order
    dy, dg2, db2 = H.bn_relu_pool_bwd(yd, dd, coef, False)
    dw2 = H.conv_first_wgrad(x.to(DEV), dy)
    assert rel(dw.cpu().numpy(), dw2.cpu().numpy()) < 1e-5 and torch.equal(dg, dg2) and torch.equal(db, db2)


def test_config1_run_spatialstream_on_hip():
    """BASELINE config 1 through the HIP path: VGG (3-conv-at-14 decoder) + plumbing + late_fusion(out, weighted)."""
    from egaze_amd.run_spatialstream import VGG, predict
    from egaze_amd.models.late_fusion import late_fusion
    from egaze_amd.utils import make_layers, cfg
    gold = np.load(os.path.join(GOLDEN, "config1.npz"))
    model = VGG(make_layers(cfg['D'], 3))
    assert list(model.state_dict()) == list(O.spatial_vgg_shapes())
    model.load_state_dict(synth.synth_state_dict(O.spatial_vgg_shapes(), seed=4, head_gain=0.25))
    model.to(DEV).eval()
    lf = build().eval()
    im_u8 = np.random.RandomState(21).randint(0, 256, (224, 224, 3)).astype(np.uint8)
    r = predict(model, lf, im_u8, DEV)
    assert rel(r["out"].cpu().numpy(), gold["out"]) < 1e-4
    assert rel(r["feat"].double().sum(dim=(2, 3)).cpu().numpy(), gold["feat_sum"]) < 1e-4
    assert np.abs(r["imq"].astype(int) - gold["imq"].astype(int)).max() <= 1        # uint8 truncation boundary
    assert np.allclose(r["predicted"], gold["predicted"], atol=0.05)
    assert rel(r["vec"].cpu().numpy(), gold["vec"]) < 1e-4
    assert rel(r["weighted"].cpu().numpy(), gold["weighted"]) < 1e-4
    assert rel(r["fin"].cpu().numpy(), gold["fin"]) < 1e-4
    # the same iteration with the batch-1 forward replayed as one captured hipGraph (--hipgraph): identical stages
    from egaze_amd.graphs import GraphedModule
    g = GraphedModule(model, (torch.zeros(1, 3, 224, 224, device=DEV),))
    r2 = predict(model, lf, im_u8, DEV, g)
    for k in ("out", "feat", "weighted", "fin"):
        assert torch.equal(r2[k], r[k]), k
    assert np.array_equal(r2["imq"], r["imq"])


def test_config1_glue_on_device_matches_host_glue_and_golden():
    """The per-frame loop body of run_spatialstream.py:125-138 with its glue on the device (SpatialPipeline: uint8 centre of
    mass, crop mean, weighted min-max, bilinear x16, no cat kernel) vs the reference golden and vs ``predict`` (host glue:
    scipy + torch), eager and as ONE captured hipGraph."""
    from egaze_amd.run_spatialstream import VGG, SpatialPipeline, predict, predict_device
    from egaze_amd.graphs import GraphedModule
    from egaze_amd.utils import make_layers, cfg
    gold = np.load(os.path.join(GOLDEN, "config1.npz"))
    model = VGG(make_layers(cfg['D'], 3))
    model.load_state_dict(synth.synth_state_dict(O.spatial_vgg_shapes(), seed=4, head_gain=0.25))
    model.to(DEV).eval()
    lf = build().eval()
    im_u8 = np.random.RandomState(21).randint(0, 256, (224, 224, 3)).astype(np.uint8)
    host = predict(model, lf, im_u8, DEV)
    pipe = SpatialPipeline(model, lf).eval()
    r = predict_device(pipe, im_u8, DEV)
    assert np.array_equal(r["predicted"], host["predicted"])            # scipy's float64 centre of mass, bit for bit
    assert np.allclose(r["predicted"], gold["predicted"], atol=0.05)
    assert torch.equal(r["out"], host["out"])
    assert rel(r["vec"].cpu().numpy(), gold["vec"]) < 1e-4
    assert rel(r["weighted"].cpu().numpy(), gold["weighted"]) < 1e-4
    assert rel(r["weighted"].cpu().numpy(), host["weighted"].cpu().numpy()) < 1e-5
    assert rel(r["fin"].cpu().numpy(), gold["fin"]) < 1e-4
    g = GraphedModule(pipe, (torch.zeros(1, 3, 224, 224, device=DEV),))
    r2 = predict_device(g, im_u8, DEV)
    for k in ("out", "weighted", "fin", "vec"):
        assert torch.equal(r2[k], r[k]), k
    assert np.array_equal(r2["predicted"], r["predicted"])
    im2 = np.random.RandomState(22).randint(0, 256, (224, 224, 3)).astype(np.uint8)     # the replay follows its input
    r3, h3 = predict_device(g, im2, DEV), predict(model, lf, im2, DEV)
    assert np.array_equal(r3["predicted"], h3["predicted"]) and rel(r3["fin"].cpu().numpy(), h3["fin"].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("align", [False, True])
def test_bilinear_up_and_u8_center_of_mass(align):
    import egaze_amd.hipops as H
    from scipy import ndimage
    g = torch.Generator().manual_seed(3)
    src = torch.rand(3, 14, 14, generator=g)
    ref = torch.nn.functional.interpolate(src.unsqueeze(1), scale_factor=16, mode="bilinear", align_corners=align).squeeze(1)
    wide = torch.zeros(3, 2, 224, 224, device=DEV)
    got = H.bilinear_up(src.to(DEV), 16, align_corners=align, out=wide[:, 1])
    assert rel(got.cpu().numpy(), ref.numpy()) < 2e-6 and float(wide[:, 0].abs().max()) == 0.0
    maps = torch.rand(3, 224, 224, generator=g)
    maps[1] *= 0.02                                                # nearly black image: tiny integer sums (levels 0 .. 5)
    com, gp, q = H.u8_center_of_mass(maps.to(DEV), want_u8=True)
    for b in range(3):
        imq = (maps[b].numpy() * 255).astype(np.uint8)
        assert np.array_equal(q[b].cpu().numpy(), imq)
        want = np.array(ndimage.center_of_mass(imq))
        assert np.array_equal(com[b].cpu().numpy(), want), (com[b], want)
        assert np.array_equal(gp[b].cpu().numpy(), np.floor(want).astype(np.int32))


@pytest.mark.parametrize("table_miss", [False, True])
def test_lf_train_step_graphed_matches_eager(table_miss):
    """LF.trainLate's iteration (LF.py:90-100) captured into one hipGraph (graphs.GraphedTrainStep) vs the same steps issued
    launch by launch: identical losses, outputs, parameters, BN running statistics and Adam state after five steps -- the
    capture changes how the step is issued (every weight-gradient fork joins back), not what it computes.
    ``table_miss``: the one-launch repack's pointer table for the step's packings is NOT cached when the capture
    starts (another packing set was touched since the warm-up steps / the table cache was evicted) -- the captured step must
    still repack after its Adam update on every replay (it falls back to the capturable per-packing launches)."""
    import egaze_amd.hipops as H
    from egaze_amd.floss import floss
    from egaze_amd.graphs import GraphedTrainStep
    from egaze_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(11)
    batches = [tuple(torch.rand(4, 1, 224, 224, generator=g).to(DEV) for _ in range(3)) for _ in range(5)]
    results = []
    for graphed in (False, True):
        torch.manual_seed(5)
        model = build()
        model.train()
        crit = floss().to(DEV)
        opt = FusedAdam(model.parameters(), lr=1e-3)
        losses = []
        if graphed:
            def fwd_loss(f, i, t):
                o = model(f, i)
                return crit(o, t), o
            step = GraphedTrainStep(fwd_loss, opt, batches[0], warm=2)
            for n, (f, i, t) in enumerate(batches):
                if n == 4:                # a parameter overwritten from outside between two REPLAYS: the replay must see it
                    with torch.no_grad():
                        model.fusion[3].weight.mul_(0.5)
                if n == 2 and table_miss:           # the call that captures finds no table for its packing set
                    H._BATCH_TABLES.clear()
                    H._BATCH_PINNED.clear()
                l, o = step(f, i, t)
                losses.append(l.item())
            assert step.graph is not None and opt.step_count == 5
            step.close()
            assert opt.step_count == 5
        else:
            for n, (f, i, t) in enumerate(batches):
                if n == 4:
                    with torch.no_grad():
                        model.fusion[3].weight.mul_(0.5)
                o = model(f, i)
                l = crit(o, t)
                opt.zero_grad()
                l.backward()
                opt.step()
                losses.append(l.item())
        torch.cuda.synchronize()
        results.append((losses, o.detach().clone(), opt.flat_p.clone(), opt.flat_m.clone(), opt.flat_v.clone(),
                        {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k}))
    (l0, o0, p0, m0, v0, b0), (l1, o1, p1, m1, v1, b1) = results
    assert l0 == l1, (l0, l1)
    assert torch.equal(o0, o1) and torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


def test_cat2_planes_matches_torch_cat():
    """late_fusion's input concatenation (models/late_fusion.py:19) as one kernel: bit-identical to torch.cat, and the model
    takes that route for contiguous device maps."""
    import egaze_amd.hipops as H
    g = torch.Generator().manual_seed(5)
    f = torch.rand(3, 1, 20, 12, generator=g).to(DEV)
    w = torch.rand(3, 1, 20, 12, generator=g).to(DEV)
    assert torch.equal(H.cat2_planes(f, w), torch.cat((f, w), dim=1))
    # every device input takes the kernel (no stock-torch branch left in late_fusion.forward): planes that are not
    # a multiple of four floats, operands that are not 16-byte aligned, and inputs that require grad (the gradient of a
    # concatenation = the two halves of the incoming gradient)
    from egaze_amd.functions import Cat2Planes
    odd = torch.rand(2 * 7 * 9 + 1, generator=g).to(DEV)
    fo, wo = odd[1:].view(2, 1, 7, 9), odd[:-1].view(2, 1, 7, 9)                    # fo is 4-byte aligned only
    assert torch.equal(Cat2Planes.apply(fo, wo), torch.cat((fo, wo), dim=1))
    fr, wr = f.clone().requires_grad_(True), w.clone().requires_grad_(True)
    out = Cat2Planes.apply(fr, wr)
    seed = torch.rand(out.shape, generator=g).to(DEV)
    out.backward(seed)
    assert torch.equal(fr.grad, seed[:, 0:1]) and torch.equal(wr.grad, seed[:, 1:2])
    with pytest.raises(RuntimeError):
        Cat2Planes.apply(f, w[:, :, :10])


def test_lf_epoch_trailing_partial_batch_and_interrupted_epoch(monkeypatch):
    """LF._run with the captured step: an epoch of full batches followed by a PARTIAL last batch (which takes the
    eager path while the optimizer is still in capturable mode) ends with the same parameters, Adam moments and step count as
    the launch-by-launch epoch, and the state dict carries the right step; an exception inside the loop still leaves the
    optimizer out of capturable mode with the host step count synced."""
    import egaze_amd.LF as lf_mod
    from egaze_amd.floss import floss
    from egaze_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(21)

    def batch(n):
        return {k: torch.rand(n, 1, 224, 224, generator=g) for k in ('im', 'gt', 'feat')}
    loader = [batch(4) for _ in range(5)] + [batch(2)]

    def shell():
        torch.manual_seed(7)
        s = object.__new__(lf_mod.LF)
        s.model = build()
        s.model.train()
        s.device = torch.device(DEV)
        s.criterion = floss().to(DEV)
        s.optimizer = FusedAdam(s.model.parameters(), lr=1e-3)
        s.epochnow = 0
        return s
    res = []
    for graphed in (False, True):
        monkeypatch.setattr(lf_mod, "LF_GRAPH", graphed)
        s = shell()
        loss, auc, aae = s._run(loader, True, 3 if graphed else 10 ** 9)       # (graphed: a print + drain every 3 iterations)
        torch.cuda.synchronize()
        assert not s.optimizer.capturable and s.optimizer.step_count == 6
        sd = s.optimizer.state_dict()
        assert float(sd["state"][0]["step"]) == 6.0
        res.append(((loss, auc, aae), s.optimizer.flat_p.clone(), s.optimizer.flat_m.clone(), s.optimizer.flat_v.clone()))
    # the captured iteration carries the batch metric inside the graph and parks loss / AAE / AUC for a later read-back
    # (LF.GraphedLateIteration): the epoch's three averages are the same numbers as the launch-by-launch loop's
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)
    # an interrupted epoch: the fourth batch raises inside the loop (after the capture)
    monkeypatch.setattr(lf_mod, "LF_GRAPH", True)
    s = shell()

    def broken():
        for i, b in enumerate(loader):
            if i == 4:
                raise KeyboardInterrupt
            yield b

    class L:
        def __iter__(self):
            return broken()

        def __len__(self):
            return len(loader)
    with pytest.raises(KeyboardInterrupt):
        s._run(L(), True, 10 ** 9)
    torch.cuda.synchronize()
    # (the loader runs one batch ahead of the step: the interrupt arrives while step 3 or 4 is the last one issued)
    assert not s.optimizer.capturable and s.optimizer.step_count in (3, 4)
    assert int(s.optimizer.step_dev[0].item()) == s.optimizer.step_count


@pytest.mark.gpu
@pytest.mark.parametrize("size", [32, 224])
def test_late_fusion_two_product_backward(size, two_products):
    """The opt-in backward arithmetic (EGAZE_BWD_PRODUCTS=2, the `two_products` fixture) on the late-fusion stack: its narrow kernels
    (persistent narrow data gradient, the tap-packed and 32 x 32 weight-gradient kernels) stage the halo operand as ONE f16 plane.
    Against the default (three products) on the same step: the forward pass does not know the knob (bit-identical map and loss),
    every gradient tensor within 3e-3 relative L2 (one operand with 11 significant bits; observed ~3e-4), and the knob reaches the
    launches (at least one tensor differs)."""
    import egaze_amd.hipops as H
    from egaze_amd.floss import floss
    assert H.BWD_PRODUCTS == 2
    im, feat, gt = synth.synth_lf_batch(2, size, seed=11)
    res = []
    for products in (2, 3):
        H.BWD_PRODUCTS = products                      # (the fixture's monkeypatch restores the module default afterwards)
        net = build()
        net.train()
        out = net(feat.to(DEV), im.to(DEV))
        loss = floss().to(DEV)(out, gt.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), loss.item(), {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()}))
    (o2, l2, g2), (o3, l3, g3) = res
    assert torch.equal(o2, o3) and l2 == l3
    differ, worst = 0, 0.0
    for k in g3:
        n = g3[k].norm().item()
        if n == 0.0:
            continue
        e = (g2[k] - g3[k]).norm().item() / n
        worst = max(worst, e)
        differ += int(e > 0)
        assert e < 3e-3, (k, e)
    print(f"late fusion {size} x {size}: two- vs three-product gradients, worst relative L2 {worst:.2e}")
    assert differ >= 1
