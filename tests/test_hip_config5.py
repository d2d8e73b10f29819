"""BASELINE config 5 -- the hand-over between the stages, against fixtures produced by the reference's own methods
(tests/golden/make_golden.py gen_config5: AT.extract_late, AT.py:199-253, then one LF.trainLate step, LF.py:83-100):
  * the SP gaze map as the uint8 image extract_late writes (truncation of 255 * x): exact, up to +-1 LSB on the few pixels
    whose 255 * x sits within float rounding of an integer;
  * the AT-weighted conv5_3 map (14 x 14 uint8, before the resize): fixation frame (crop mean only) and the two saccade
    frames (LSTM branch with the hidden state carried over) -- within +-1 LSB;
  * one LF training step on the reference's extracted maps: output map, floss, AAE / AUC, parameters after Adam."""
import os

import numpy as np
import pytest
import torch

from oracle import egaze_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "config5.npz")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_extract_late_matches_reference(tmp_path, monkeypatch):
    from PIL import Image
    import egaze_amd  # noqa: F401
    import egaze_amd.AT as at_mod
    from egaze_amd.AT import AT
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.utils import cfg, make_layers
    gold = np.load(GOLD)
    save = tmp_path / "save"
    save.mkdir()
    torch.save({'state_dict': synth.synth_state_dict(O.sp_shapes(), seed=1, head_gain=0.25)}, str(save / "sp.pth.tar"))
    torch.save(synth.synth_state_dict(O.lstm_shapes(), seed=2), str(save / "lstm.pth.tar"))
    for sub in ("train", "test"):                      # AT's constructor lists the LSTM training folders
        d = tmp_path / "512w" / sub
        d.mkdir(parents=True)
        for i in range(2):
            torch.save(torch.zeros(512), str(d / f"fix_v_{i:010d}.pth.tar"))
    at = AT(pretrained_model=str(save / "sp.pth.tar"), pretrained_lstm=str(save / "lstm.pth.tar"), save_path=str(save),
            device='0', lstm_data_path=str(tmp_path / "512w"))
    n = 3
    x_s, x_t, gt, _ = synth.synth_sp_batch(n, 224, seed=31)
    fixsac = gold["fixsac"]
    loader = [{"imname": ["f%d.png" % i], "fixsac": torch.tensor([[float(fixsac[i])]]), "image": x_s[i:i + 1],
               "flow": x_t[i:i + 1], "gt": gt[i:i + 1]} for i in range(n)]
    small = []
    real_resize = at_mod.resize
    monkeypatch.setattr(at_mod, "resize", lambda arr, size: (small.append(np.array(arr, copy=True)), real_resize(arr, size))[1])
    pred_dir, feat_dir = str(tmp_path / "pred") + "/", str(tmp_path / "feat") + "/"
    at.extract_late(loader, pred_dir, feat_dir)
    pred = np.stack([np.asarray(Image.open(os.path.join(pred_dir, "f%d.png" % i))) for i in range(n)])
    d = np.abs(pred.astype(np.int32) - gold["pred_u8"].astype(np.int32))
    print("pred uint8: max |diff|", d.max(), "pixels off by one:", int((d == 1).sum()), "of", d.size)
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    feat14 = np.stack(small)
    assert feat14.shape == gold["feat14_u8"].shape and feat14.dtype == np.uint8
    d14 = np.abs(feat14.astype(np.int32) - gold["feat14_u8"].astype(np.int32))
    print("AT map 14x14 uint8: max |diff|", d14.max(), "cells off by one:", int((d14 == 1).sum()), "of", d14.size)
    assert d14.max() <= 1 and (d14 > 0).mean() < 0.02
    feat = np.stack([np.asarray(Image.open(os.path.join(feat_dir, "f%d.png" % i))) for i in range(n)])
    assert feat.shape == (n, 224, 224)
    # the resize is host I/O (cv2 in the reference, cv2 / PIL here): same map up to the interpolation's own rounding
    assert np.abs(feat.astype(np.int32) - gold["feat_u8"].astype(np.int32)).max() <= 3


def test_lf_step_on_extracted_maps():
    import egaze_amd  # noqa: F401
    from egaze_amd.floss import floss
    from egaze_amd.models.late_fusion import late_fusion
    from egaze_amd.optim import FusedAdam
    from egaze_amd.utils import computeAAEAUC
    gold = np.load(GOLD)
    dev = "cuda:0"
    lf = late_fusion()
    lf.load_state_dict(synth.synth_state_dict(O.lf_shapes(), seed=3, head_gain=0.5))
    lf.to(dev)
    _, _, gt, _ = synth.synth_sp_batch(3, 224, seed=31)
    im = torch.from_numpy(gold["pred_u8"]).float().div(255).unsqueeze(1).to(dev)
    ft = torch.from_numpy(gold["feat_u8"]).float().div(255).unsqueeze(1).to(dev)
    gtq = torch.from_numpy(np.uint8(np.round(gt.numpy() * 255))).float().div(255).to(dev)
    opt = FusedAdam(lf.parameters(), lr=1e-4)
    out = lf(ft, im)                                                    # channel 0 = AT map, 1 = SP map (LF.py:90)
    loss = floss()(out, gtq)
    aae, auc, _ = computeAAEAUC(out.detach(), gtq)
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert rel(out.detach().cpu().numpy(), gold["lf_out"]) < 1e-4
    assert abs(loss.item() - float(gold["lf_loss"])) < 1e-5 * abs(float(gold["lf_loss"]))
    assert abs(aae - gold["lf_aae_auc"][0]) < 1e-3 and abs(auc - gold["lf_aae_auc"][1]) < 1e-6
    # Parameters after the Adam step.  Step 1 of Adam moves every element by -lr * g / (|g| + eps) ~ -lr * sign(g), so the
    # change of a tensor's sum counts gradient signs: at most 2 % of the elements may disagree with the reference.  The
    # conv biases in front of a train-mode BatchNorm are excluded: their true gradient is zero, the reference moves them by
    # its fp32 round-off noise, this package leaves them alone (functions._zero_bias_grad).
    before = synth.synth_state_dict(O.lf_shapes(), seed=3, head_gain=0.5)
    for k, p in lf.named_parameters():
        if k in ("fusion.0.bias", "fusion.3.bias", "fusion.6.bias"):
            assert torch.equal(p.detach().cpu(), before[k])
            continue
        got = np.array([p.detach().double().sum().item(), p.detach().double().norm().item()])
        want = gold["lf_after_sum/" + k]
        moved = abs(want[0] - before[k].double().sum().item())
        assert abs(got[0] - want[0]) <= 2e-4 * (0.02 * p.numel() + 1), (k, got, want, moved)
        assert abs(got[1] - want[1]) <= 1e-4 * want[1] + 1e-6, (k, got, want)
