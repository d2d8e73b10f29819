"""GPU parity of the device computeAAEAUC (csrc/metrics.hip) with the reference's host metric (utils.py:96-140):
the golden fixture generated from the reference, the scipy-based oracle on border / tie cases, and the drivers' entry."""
import os

import numpy as np
import pytest
import torch

from oracle import egaze_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def test_aae_auc_golden_fixture():
    from egaze_amd.utils import computeAAEAUC
    gold = np.load(os.path.join(GOLDEN, "metrics_glue.npz"))
    rs = np.random.RandomState(11)
    gt = synth.synth_gt(3, 224, rs)[:, 0]
    pred = synth.synth_gt(3, 224, rs)[:, 0] * 0.8 + rs.uniform(0, 0.05, (3, 224, 224)).astype(np.float32)
    aae, auc, gp = computeAAEAUC(_dev(pred), _dev(gt))                       # batch branch (utils.py:100-121)
    assert abs(aae - gold["batch_aae_auc"][0]) < 1e-6 and abs(auc - gold["batch_aae_auc"][1]) < 1e-12
    assert np.array_equal(np.array(gp), gold["batch_gp"])
    aae1, auc1, gp1 = computeAAEAUC(_dev(pred[1]), _dev(gt[1]))              # single-image branch (:122-140)
    assert abs(aae1 - gold["single_aae_auc"][0]) < 1e-6 and abs(auc1 - gold["single_aae_auc"][1]) < 1e-12
    assert np.array_equal(np.array(gp1), gold["single_gp"])
    # the (B,1,H,W) tensors the drivers hand over squeeze to the same thing
    aae2, auc2, _ = computeAAEAUC(_dev(pred)[:, None], _dev(gt)[:, None])
    assert aae2 == aae and auc2 == auc


def test_aae_auc_borders_and_ties_vs_oracle():
    """Centroids whose filtered delta is reflected at the image border (direct + mirrored taps overlap), the flat and
    multi-peak targets (first arg-max), and an exact count check: fp is an integer and must be identical."""
    import egaze_amd.hipops as H
    rs = np.random.RandomState(3)
    preds, gts = [], []
    for (ci, cj) in [(0, 0), (223, 223), (3, 219), (40, 60), (111, 112), (55, 56), (57, 166), (200, 10), (1, 100)]:
        p = np.full((224, 224), 1e-6, np.float32)
        p[max(ci - 1, 0):ci + 2, max(cj - 1, 0):cj + 2] += rs.uniform(0.5, 1.0)
        preds.append(p)
        gts.append(synth.synth_gt(1, 224, rs)[0, 0])
    flat = np.full((224, 224), 0.25, np.float32)                              # all-equal target: arg-max = (0, 0)
    two = np.zeros((224, 224), np.float32); two[200, 30] = 0.7; two[10, 20] = 0.7   # ties: first in row-major order
    preds += [synth.synth_gt(1, 224, rs)[0, 0], synth.synth_gt(1, 224, rs)[0, 0]]
    gts += [flat, two]
    pred, gt = np.stack(preds), np.stack(gts)
    res = H.aae_auc(_dev(pred), _dev(gt)).cpu().numpy()
    for b in range(len(preds)):
        a, auc, gp = O.compute_aae_auc(pred[b], gt[b])
        fp_ref = round((1 - auc) * 224 * 224)
        assert [int(res[b, 2]), int(res[b, 3])] == list(gp[0]), b
        assert int(res[b, 1]) == fp_ref, (b, res[b], fp_ref)
        # the reference divides by a float32 numpy sum (1 ulp = 8e-8 relative): up to ~2e-5 px of centroid, ~2e-6 deg
        assert abs(res[b, 0] - a) < 1e-5, (b, res[b, 0], a)
    assert [int(res[-2, 2]), int(res[-2, 3])] == [0, 0] and [int(res[-1, 2]), int(res[-1, 3])] == [10, 20]


def test_aae_auc_rejects_other_sizes():
    import egaze_amd.hipops as H
    from egaze_amd._lib import EgazeHipError
    x = torch.rand(2, 112, 112, device=DEV)
    with pytest.raises(EgazeHipError):
        H.aae_auc(x, x)


def test_u8_normalize_bit_exact():
    """Device input pipeline (data/STdatas.py:50-68): same three fp32 operations as the reference's torch expression."""
    import egaze_amd.hipops as H
    from egaze_amd.data.STdatas import IMAGE_MEAN, IMAGE_STD, FLOW_MEAN, FLOW_STD, stage_batch
    rs = np.random.RandomState(4)
    im = torch.from_numpy(rs.randint(0, 256, (3, 3, 32, 20)).astype(np.uint8))
    fl = torch.from_numpy(rs.randint(0, 256, (3, 20, 32, 20)).astype(np.uint8))
    gt = torch.from_numpy(rs.randint(0, 256, (3, 1, 32, 20)).astype(np.uint8))
    mean = torch.tensor(IMAGE_MEAN).view(3, 1, 1); std = torch.tensor(IMAGE_STD).view(3, 1, 1)
    ref_im = (im.float().div(255) - mean) / std
    ref_fl = (fl.float().div(255) - 0.5) / 0.5
    ref_gt = gt.float().div(255)
    a, b, c = stage_batch({'image': im, 'flow': fl, 'gt': gt}, DEV)
    assert torch.equal(a.cpu(), ref_im) and torch.equal(b.cpu(), ref_fl) and torch.equal(c.cpu(), ref_gt)
    a2, b2, c2 = stage_batch({'image': ref_im, 'flow': ref_fl, 'gt': ref_gt}, DEV)       # already-normalised samples pass through
    assert torch.equal(a2.cpu(), ref_im) and torch.equal(c2.cpu(), ref_gt)
    with pytest.raises(ValueError):
        H.u8_normalize(im.to(DEV), (0.5,), (0.5,))


def test_staged_batches_prefetch_order_and_values():
    """data.STdatas.staged_batches (batch k + 1 copied on a copy stream while step k computes): every batch arrives, in
    order, bit-identical to in-step staging, for raw-byte and fp32 loaders, while the consumer stream is kept busy."""
    from egaze_amd.data.STdatas import stage_batch, staged_batches
    rs = np.random.RandomState(9)
    for raw in (True, False):
        batches = []
        for k in range(5):
            im = torch.from_numpy(rs.randint(0, 256, (2, 3, 32, 32)).astype(np.uint8))
            fl = torch.from_numpy(rs.randint(0, 256, (2, 20, 32, 32)).astype(np.uint8))
            gt = torch.from_numpy(rs.randint(0, 256, (2, 1, 32, 32)).astype(np.uint8))
            smp = {'image': im, 'flow': fl, 'gt': gt, 'k': k}
            if not raw:
                smp = {'image': im.float(), 'flow': fl.float(), 'gt': gt.float(), 'k': k}
            batches.append({n: (t.pin_memory() if torch.is_tensor(t) else t) for n, t in smp.items()})
        busy = torch.randn(2048, 2048, device=DEV)
        seen = 0
        for k, (sample, staged) in enumerate(staged_batches(batches, DEV)):
            assert sample['k'] == k
            busy = busy @ busy.clamp(-1e-3, 1e-3)                  # consumer-stream work the next copy overlaps with
            sums = [t.double().sum() for t in staged]              # consume on the current stream
            ref = stage_batch(batches[k], DEV)
            for t, r, s_ in zip(staged, ref, sums):
                assert torch.equal(t, r) and s_.item() == r.double().sum().item()
            seen += 1
        assert seen == 5
    assert list(staged_batches([], DEV)) == []


def test_at_glue_kernels_vs_oracle():
    """AT.crop_feature + spatial mean and AT.get_weighted (AT.py:25-39,58-66,229) on the device vs the oracle."""
    from egaze_amd.AT import crop_mean_weight, get_weighted
    gold = np.load(os.path.join(GOLDEN, "metrics_glue.npz"))
    krs = np.random.RandomState(12)
    krs.standard_normal((64, 3, 3, 3)); [krs.standard_normal((4,)) for _ in range(1, 30)]     # same stream as the fixture
    feat = torch.from_numpy(np.abs(krs.standard_normal((2, 512, 14, 14))).astype(np.float32))
    gp = [[5, 220], [117, 60]]
    fd = feat.to(DEV).contiguous(memory_format=torch.channels_last)
    w = crop_mean_weight(fd, gp, 3)
    ref_crop = torch.from_numpy(gold["crop_feature"])
    ref_w = ref_crop.contiguous().view(2, 512, -1).mean(2)
    assert (w.cpu() - ref_w).abs().max().item() < 1e-6 * ref_w.abs().max().item()
    for size, pts in [(1, [[0, 0], [223, 223]]), (2, [[100, 37], [16, 208]]), (5, [[3, 3], [222, 111]])]:
        ow = O.crop_feature(feat, pts, size).contiguous().view(2, 512, -1).mean(2)
        assert (crop_mean_weight(fd, pts, size).cpu() - ow).abs().max().item() < 1e-6 * ow.abs().max().item()
    got = get_weighted(w[0], fd[0:1])
    assert tuple(got.shape) == (1, 14, 14)
    ref = gold["get_weighted"]
    assert np.abs(got.cpu().numpy() - ref).max() < 2e-6
    assert float(got.min()) == 0.0 and float(got.max()) == 1.0


def test_window_and_align_means_on_device():
    """egz_window_mean (extractLSTMw.crop_feature_var + mean) and egz_pixel_weighted_sum (the --align crop mean) against the
    host formulations on a (B, 512, 14, 14) map."""
    import egaze_amd  # noqa: F401
    from egaze_amd import extractLSTMw as ex
    from egaze_amd.AT import crop_align_feature, crop_align_mean
    rs = np.random.RandomState(3)
    feat = torch.from_numpy(np.abs(rs.standard_normal((3, 512, 14, 14))).astype(np.float32))
    fd = feat.to("cuda:0").contiguous(memory_format=torch.channels_last)
    gps = [[5, 220], [117, 60], [223, 0]]
    ref = crop_align_feature(feat, gps, 3).contiguous().view(3, 512, -1).mean(2)
    got = crop_align_mean(fd, gps, 3).cpu()
    assert np.abs(got.numpy() - ref.numpy()).max() < 1e-5 * np.abs(ref.numpy()).max()
    # extractLSTMw window (float clip + int slicing) through channel_weight: device vs host path, one sample at a time
    for ind in (0, 13, 5 * 14 + 5, 195):
        gt = torch.zeros(1, 1, 224, 224)
        gt[0, 0, (ind // 14) * 16 + 8, (ind % 14) * 16 + 8] = 1.0
        w_dev = ex.channel_weight(fd[:1], gt, 3, False).cpu()
        w_cpu = ex.channel_weight(feat[:1], gt, 3, False)
        assert np.allclose(w_dev.numpy(), w_cpu.numpy(), rtol=1e-5, atol=1e-7), ind
        a_dev = ex.channel_weight(fd[:1], gt, 3, True).cpu()
        a_cpu = ex.channel_weight(feat[:1], gt, 3, True)
        assert np.allclose(a_dev.numpy(), a_cpu.numpy(), rtol=1e-4, atol=1e-6), ind


def test_glue_device_paths_equal_their_host_branches():
    """the AT / extraction glue keeps host branches (the reference's own formulations, AT.py:25-66,
    extractLSTMw.py:46-81) next to the device kernels.  Every one of them against its device path on the same values: device
    tensors never take a stock-torch route, host tensors reproduce the reference."""
    from egaze_amd import AT as at
    rs = np.random.RandomState(21)
    feat = torch.from_numpy(np.abs(rs.standard_normal((4, 512, 14, 14))).astype(np.float32))
    fd = feat.to(DEV).contiguous(memory_format=torch.channels_last)
    gps = [[5, 220], [117, 60], [223, 0], [0, 113]]
    w_host = at.crop_mean_weight(feat, gps, 3)
    w_dev = at.crop_mean_weight(fd, gps, 3)
    assert w_dev.is_cuda and np.allclose(w_dev.cpu().numpy(), w_host.numpy(), rtol=1e-6, atol=1e-7)
    gp_dev = torch.tensor(gps, dtype=torch.int32, device=DEV)
    assert torch.equal(at.crop_mean_weight(fd, gp_dev, 3), w_dev)                 # gaze points already on the device
    a_host, a_dev = at.crop_align_mean(feat, gps, 3), at.crop_align_mean(fd, gps, 3)
    assert np.abs(a_dev.cpu().numpy() - a_host.numpy()).max() < 1e-5 * np.abs(a_host.numpy()).max()
    b_host, b_dev = at.get_weighted_batch(w_host, feat), at.get_weighted_batch(w_dev, fd)
    assert tuple(b_dev.shape) == (4, 14, 14) and np.abs(b_dev.cpu().numpy() - b_host.numpy()).max() < 2e-6
    for n in range(4):
        one_host = at.get_weighted(w_host[n], feat[n:n + 1])
        one_dev = at.get_weighted(w_dev[n], fd[n:n + 1])
        assert np.abs(one_dev.cpu().numpy() - one_host.numpy()).max() < 2e-6
        assert np.abs(one_dev.cpu().numpy() - b_host[n:n + 1].numpy()).max() < 2e-6
    with pytest.raises(NotImplementedError):
        at.get_weighted(w_dev[0], fd[:2])
    with pytest.raises(NotImplementedError):
        at.crop_align_mean(fd[:, :, :, :7].contiguous(), gps, 3)
