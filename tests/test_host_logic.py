"""CPU-only tests of the host side: C-ABI symbol table vs header vs the built .so, module construction /
state-dict layout, CLI flags, host glue (change_key_names, computeAAEAUC, AverageMeter, repackage_hidden)."""
import collections
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_cabi_header_table_and_library_agree():
    import egaze_amd
    from egaze_amd import _lib
    header = open(os.path.join(ROOT, "include", "egaze_hip.h")).read()
    declared = set(re.findall(r"\b(egz_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (egz_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert "gfx950" in egaze_amd.version()


def test_state_dict_layouts_match_reference():
    from oracle import egaze_oracle as O
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.models.LSTMnet import lstmnet
    from egaze_amd.models.late_fusion import late_fusion
    from egaze_amd.utils import make_layers, cfg
    m = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20))
    sd = m.state_dict()
    assert list(sd) == list(O.sp_shapes()) and len(sd) == 215
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in O.sp_shapes().items())
    assert sum(p.numel() for p in m.parameters()) == 46529409
    assert len(list(m.decoder.children())) == 29 and len(list(m.features_s.children())) == 43
    assert list(lstmnet().state_dict()) == list(O.lstm_shapes())
    lf = late_fusion()
    assert list(lf.state_dict()) == list(O.lf_shapes()) and sum(p.numel() for p in lf.parameters()) == 12321
    # init semantics (models/model_SP.py:52-65): Conv2d fan-out normal + zero bias, BN 1/0, Conv3d untouched
    assert float(m.decoder[0].bias.abs().max()) == 0 and float(m.bn.weight.min()) == 1
    assert float(m.fusion.bias.abs().max()) > 0
    std = float(m.decoder[0].weight.std())
    assert abs(std - (2.0 / (9 * 512)) ** 0.5) < 0.05 * (2.0 / (9 * 512)) ** 0.5


def test_cpu_tensors_are_rejected():
    from egaze_amd.models.late_fusion import late_fusion
    from egaze_amd.floss import floss
    with pytest.raises(RuntimeError):
        late_fusion()(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError):
        floss()(torch.rand(1, 1, 8, 8), torch.rand(1, 1, 8, 8))


def test_cli_flags_match_reference():
    from egaze_amd.gaze_full import build_parser
    a = build_parser().parse_args([])
    assert len(vars(a)) == 37
    assert (a.lr, a.lr_late, a.batch_size, a.batch_size_sp, a.crop_size, a.val_name, a.loss_function, a.sp_resume,
            a.num_epoch, a.num_epoch_lstm, a.device) == (1e-7, 1e-4, 64, 8, 3, 'Alireza', 'f', '0', 10, 120, '0')


def test_host_glue_against_golden():
    from egaze_amd.utils import change_key_names, computeAAEAUC, AverageMeter, repackage_hidden
    from oracle import synth
    gold = np.load(os.path.join(GOLDEN, "metrics_glue.npz"))
    rs = np.random.RandomState(11)
    gt = synth.synth_gt(3, 224, rs)[:, 0]
    pred = synth.synth_gt(3, 224, rs)[:, 0] * 0.8 + rs.uniform(0, 0.05, (3, 224, 224)).astype(np.float32)
    aae, auc, gp = computeAAEAUC(pred, gt)
    assert np.allclose([aae, auc], gold["batch_aae_auc"], rtol=1e-12) and np.array_equal(np.array(gp), gold["batch_gp"])
    a1, u1, g1 = computeAAEAUC(pred[1], gt[1])
    assert np.allclose([a1, u1], gold["single_aae_auc"], rtol=1e-12) and np.array_equal(np.array(g1), gold["single_gp"])
    od = collections.OrderedDict()
    krs = np.random.RandomState(12)
    od["features.0.weight"] = torch.from_numpy(krs.standard_normal((64, 3, 3, 3)).astype(np.float32))
    for n in range(1, 30):
        od[f"features.k{n}"] = torch.from_numpy(krs.standard_normal((4,)).astype(np.float32))
    new = change_key_names(od, 20)
    assert list(new) == list(gold["ckn_keys"]) and np.array_equal(new["features.0.weight"].numpy(), gold["ckn_w0"])
    m = AverageMeter()
    m.update(2.0, 3); m.update(4.0)
    assert m.avg == 2.5 and m.count == 4
    h = (torch.ones(2, requires_grad=True) * 2, torch.ones(2, requires_grad=True) * 3)
    d = repackage_hidden(h)
    assert not d[0].requires_grad and repackage_hidden(None) is None


def test_stdataset_raw_u8_contract(tmp_path):
    """data/STdatas.py mirror on files: the raw_u8 samples (bytes for the device input pipeline) normalise on the host to
    exactly the reference-contract samples; flow window looks 10 frames back; fixsac is dilated by one frame."""
    from PIL import Image
    from egaze_amd.data.STdatas import STDataset, IMAGE_MEAN, IMAGE_STD
    rs = np.random.RandomState(0)
    folder = "Ahmad_American"
    (tmp_path / "flow" / folder).mkdir(parents=True)
    (tmp_path / "img").mkdir(); (tmp_path / "gt").mkdir(); (tmp_path / "fs").mkdir()
    for n in range(1, 13):
        for ax in "xy":
            Image.fromarray(rs.randint(0, 256, (16, 12)).astype(np.uint8)).save(str(tmp_path / "flow" / folder / f"flow_{ax}_{n:05d}.jpg"))
    names, gts = [], []
    for n in (11, 12):
        nm = f"{folder}_img_{n:05d}.png"; g = f"{folder}_gt_{n:05d}.png"
        Image.fromarray(rs.randint(0, 256, (16, 12, 3)).astype(np.uint8)).save(str(tmp_path / "img" / nm))
        Image.fromarray(rs.randint(0, 256, (16, 12)).astype(np.uint8)).save(str(tmp_path / "gt" / (folder + "_000000_" + f"{n:05d}.png")))
        names.append(nm); gts.append(folder + "_000000_" + f"{n:05d}.png")
    np.savetxt(str(tmp_path / "fs" / "a.txt"), np.array([0.0, 1.0]))
    args = (str(tmp_path / "flow"), str(tmp_path / "img"), str(tmp_path / "gt"), [folder], names, gts, ["a.txt"], str(tmp_path / "fs"))
    ref, raw = STDataset(*args), STDataset(*args, raw_u8=True)
    for i in range(2):
        a, b = ref[i], raw[i]
        assert b['image'].dtype == torch.uint8 and b['flow'].dtype == torch.uint8 and b['gt'].dtype == torch.uint8
        assert tuple(b['flow'].shape) == (20, 16, 12) and tuple(b['image'].shape) == (3, 16, 12)
        mean = torch.tensor(IMAGE_MEAN).view(3, 1, 1); std = torch.tensor(IMAGE_STD).view(3, 1, 1)
        assert torch.equal((b['image'].float().div(255) - mean) / std, a['image'])
        assert torch.equal((b['flow'].float().div(255) - 0.5) / 0.5, a['flow'])
        assert torch.equal(b['gt'].float().div(255), a['gt'])
        assert float(a['fixsac']) == 1.0                       # [0, 1] dilated by [1, 1, 1] -> [1, 1]


def test_extract_lstm_mirror_matches_reference_golden(tmp_path):
    """extractLSTMw mirror against fixtures produced by the reference's own crop_feature_var / extractw
    (tests/golden/make_golden.py gen_extract_lstm): float-clip window for every gaze cell and crop size, the
    AvgPool2d(16) arg-max cell, the fixation state machine and the stored vectors."""
    import numpy as np
    import torch
    import egaze_amd  # noqa: F401
    from egaze_amd import extractLSTMw as ex
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "extract_lstm.npz"))
    rs = np.random.RandomState(51)
    feat = torch.from_numpy(np.abs(rs.standard_normal((1, 6, 14, 14))).astype(np.float32))
    for size in (2, 3, 4):
        for ind in range(196):
            c = ex.crop_feature_var(feat, torch.tensor([[ind]]), size).contiguous()
            assert tuple(c.shape[2:]) == tuple(gold[f"var_shape_s{size}"][ind]), (size, ind)
            assert np.allclose(c.view(1, 6, -1).mean(2).numpy()[0], gold[f"var_mean_s{size}"][ind], rtol=1e-6, atol=1e-7)
    assert ex.var_window(0, 3, 14, 14) == (0, 3, 0, 3) and ex.var_window(5 * 14 + 5, 3, 14, 14) == (3, 7, 3, 7)
    # state machine + stored vectors, driven exactly like the generator drove the reference
    from oracle import synth
    fix = [int(v) for v in gold["extractw_fix"]]
    gts = synth.synth_gt(len(fix), 224, np.random.RandomState(52))
    ims = rs.standard_normal((len(fix), 3, 224, 224)).astype(np.float32)
    loader = [{"fixsac": torch.tensor([[float(f)]]), "imname": ["vid_%05d.jpg" % i],
               "image": torch.from_numpy(ims[i:i + 1]), "gt": torch.from_numpy(gts[i:i + 1])} for i, f in enumerate(fix)]

    class Fake(torch.nn.Module):
        def forward(self, x):
            p = torch.nn.functional.avg_pool2d(x, 16)
            k = torch.arange(512, dtype=torch.float32).view(1, 512, 1, 1)
            return torch.relu(p[:, 0:1] * torch.sin(k * 0.37) + p[:, 1:2] * torch.cos(k * 0.11) + p[:, 2:3] * 0.5)

    ex.extractw(loader, Fake(), str(tmp_path), crop_size=3, device="cpu", align=False)
    names = sorted(os.listdir(str(tmp_path)))
    assert names == [str(n) for n in gold["extractw_names"]]
    vecs = np.stack([torch.load(os.path.join(str(tmp_path), n)).numpy() for n in names])
    assert np.allclose(vecs, gold["extractw_vecs"], rtol=1e-5, atol=1e-7)
    # a fixation that ends after one frame is an error in the reference (extractLSTMw.py:110-111)
    bad = [dict(loader[0], fixsac=torch.tensor([[1.0]])), dict(loader[1], fixsac=torch.tensor([[0.0]]))]
    with pytest.raises(RuntimeError):
        ex.extractw(bad, Fake(), str(tmp_path / "bad"), crop_size=3, device="cpu")


def test_align_window_weights_equal_interpolate_crop_mean():
    """`--align` chn_weight (AT.py:41-56 + :229): the mean of a window of the x16 bilinear upsampling written as per-cell
    weights on the 14 x 14 map, against torch's interpolate + crop + mean (the reference's own sequence of ops)."""
    import numpy as np
    import torch
    import egaze_amd  # noqa: F401
    from egaze_amd.AT import align_window_weights, crop_align_feature
    rs = np.random.RandomState(0)
    feat = torch.from_numpy(rs.standard_normal((1, 7, 14, 14)).astype(np.float32))
    for size in (1, 3, 5):
        for gp in ([5, 220], [117, 60], [0, 0], [223, 223], [100, 30], [111, 112]):
            ref = crop_align_feature(feat, [gp], size).contiguous().view(1, 7, -1).mean(2)[0].numpy()
            W = align_window_weights(gp, size)
            assert abs(W.sum() - 1.0) < 1e-12
            got = (feat[0].double().numpy() * W[None]).sum((1, 2))
            assert np.abs(got - ref).max() < 1e-6, (size, gp)


def test_staged_batches_cpu_fallback_keeps_order_and_values():
    """data.STdatas.staged_batches without a GPU (or with EGAZE_STREAMS=0): plain in-loop staging, every batch once, in
    order, through the caller's staging function when one is given (the LF loop's)."""
    import torch
    from egaze_amd.data.STdatas import staged_batches
    batches = [{'image': torch.full((1, 3, 4, 4), float(k)), 'flow': torch.full((1, 20, 4, 4), float(k)),
                'gt': torch.full((1, 1, 4, 4), float(k)), 'k': k} for k in range(4)]
    seen = []
    for sample, (im, fl, gt) in staged_batches(batches, 'cpu'):
        assert float(im.mean()) == float(fl.mean()) == float(gt.mean()) == sample['k']
        seen.append(sample['k'])
    assert seen == [0, 1, 2, 3]
    out = list(staged_batches(batches, 'cpu', stage=lambda s, d: (s['gt'] + 1,)))
    assert [float(t[0].mean()) for _, t in out] == [1.0, 2.0, 3.0, 4.0]
    assert list(staged_batches([], 'cpu')) == []
