"""GPU tests of the driver mirrors (SP / AT / LF classes, the reference's module API) on tiny synthetic datasets
written to a temp dir in the reference's on-disk formats."""
import collections
import os

import numpy as np
import pytest
import torch
from torch.utils.data import Dataset

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _SPData(Dataset):
    def __init__(self, n, size, seed):
        from oracle import synth
        self.im, self.fl, self.gt, self.fs = synth.synth_sp_batch(n, size, seed=seed)

    def __len__(self):
        return self.im.shape[0]

    def __getitem__(self, i):
        return {'image': self.im[i], 'flow': self.fl[i], 'gt': self.gt[i], 'fixsac': self.fs[i],
                'imname': 'frame_%05d.jpg' % i}


def _fake_vgg(path):
    """vgg16_bn-shaped state dict: 'features.<idx>.*' (13 conv + 13 BN) + classifier entries."""
    from egaze_amd.utils import make_layers, cfg
    torch.manual_seed(3)
    enc = make_layers(cfg['D'], 3)
    sd = collections.OrderedDict(('features.' + k, v.clone().normal_(0, 0.05) if v.is_floating_point() else v.clone())
                                 for k, v in enc.state_dict().items())
    for k in list(sd):
        if k.endswith('running_var'):
            sd[k] = sd[k].abs() + 0.5
    sd['classifier.0.weight'] = torch.zeros(4, 4)
    torch.save(sd, path)
    return sd


def test_sp_driver_resume0_train_val_checkpoint(tmp_path, monkeypatch):
    from egaze_amd.SP import SP
    vgg = _fake_vgg(str(tmp_path / "vgg.pth"))
    monkeypatch.setenv("EGAZE_VGG16_BN", str(tmp_path / "vgg.pth"))
    save = str(tmp_path / "save")
    sp = SP(lr=1e-4, save_path=save, save_name='best_SP.pth.tar', num_epoch=1, batch_size=2, device='0',
            resume='0', traindata=_SPData(4, 32, 0), valdata=_SPData(2, 32, 1))
    w_t0 = sp.model.features_t[0].weight.detach().cpu()
    rgb = vgg['features.0.weight']
    assert torch.allclose(w_t0, rgb.mean(1, keepdim=True).repeat(1, 20, 1, 1))        # utils.change_key_names
    assert torch.equal(sp.model.features_s[0].weight.detach().cpu(), rgb)
    assert torch.equal(sp.model.features_s[40].weight.detach().cpu(), vgg['features.40.weight'])
    # the flow stream only receives the first 25 entries (SP.py:56 + utils.py:82): later layers keep their init
    assert not torch.equal(sp.model.features_t[40].weight.detach().cpu(), vgg['features.40.weight'])
    before = sp.model.decoder[0].weight.detach().clone()
    sp.train()
    assert not torch.equal(before, sp.model.decoder[0].weight.detach())
    ck = torch.load(os.path.join(save, 'best_SP.pth.tar'), map_location='cpu', weights_only=False)
    assert set(ck) == {'epoch', 'arch', 'state_dict', 'optimizer', 'auc', 'aae'} and ck['arch'] == 'SP'
    assert len(ck['state_dict']) == 215
    # resume '2' builds the fusion+bn+decoder optimizer (SP.py:109-111): an all-parameter optimizer state does
    # not fit it -- torch raises ValueError there, so do we
    with pytest.raises(ValueError):
        SP(lr=1e-4, save_path=save, save_name='best_SP.pth.tar', num_epoch=2, batch_size=2, device='0',
           resume='2', traindata=_SPData(2, 32, 0), valdata=_SPData(2, 32, 1))
    # resume '1'-style: encoders frozen, only fusion + bn + decoder in the optimizer
    torch.save({'state_dict': {}}, str(tmp_path / "s.pth"))
    sp3 = SP(lr=1e-4, save_path=save, save_name='frozen_SP.pth.tar', num_epoch=1, batch_size=2, device='0', resume=1,
             pretrained_spatial=str(tmp_path / "s.pth"), pretrained_temporal=str(tmp_path / "s.pth"),
             traindata=_SPData(2, 32, 0), valdata=_SPData(2, 32, 1))
    assert sum(p.numel() for p in sp3.optimizer.params) == 2359808 + 1024 + 14712513
    enc_before = sp3.model.features_s[0].weight.detach().clone()
    sp3.train()
    assert torch.equal(enc_before, sp3.model.features_s[0].weight.detach())
    ck3 = torch.load(os.path.join(save, 'frozen_SP.pth.tar'), map_location='cpu', weights_only=False)
    # resume '2' from that checkpoint restores weights, epoch and optimizer moments
    sp2 = SP(lr=1e-4, save_path=save, save_name='frozen_SP.pth.tar', num_epoch=2, batch_size=2, device='0',
             resume='2', traindata=_SPData(2, 32, 0), valdata=_SPData(2, 32, 1))
    assert torch.equal(sp2.model.decoder[0].weight.detach().cpu(), ck3['state_dict']['decoder.0.weight'])
    assert sp2.optimizer.step_count == 1 and sp2.epochnow == 0
    m0 = ck3['optimizer']['state'][0]['exp_avg']
    assert torch.equal(sp2.optimizer.flat_m[:m0.numel()].cpu().view(m0.shape), m0)


def test_at_and_lf_drivers(tmp_path):
    from PIL import Image
    from egaze_amd.AT import AT, crop_feature, get_weighted
    from egaze_amd.LF import LF
    from egaze_amd.models.model_SP import model_SP
    from egaze_amd.utils import make_layers, cfg, generalException
    from oracle import synth
    with pytest.raises(generalException):
        AT(pretrained_model=None)
    save = tmp_path / "save"
    save.mkdir()
    torch.manual_seed(0)
    sp = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20))
    torch.save({'state_dict': sp.state_dict()}, str(save / "sp.pth.tar"))
    # AT training data: consecutive 512-vectors, two videos
    for sub in ("train", "test"):
        d = tmp_path / "512w" / sub
        d.mkdir(parents=True)
        ins, _ = synth.synth_at_batch(6, 1, seed=1)
        for i in range(6):
            vid = "Ahmad_Pizza1" if i < 4 else "Carlos_Tea2"
            torch.save(ins[i, 0].clone(), str(d / f"fix_{vid}_{i:010d}.pth.tar"))
    at = AT(pretrained_model=str(save / "sp.pth.tar"), num_epoch_lstm=1, save_path=str(save), device='0',
            lstm_data_path=str(tmp_path / "512w"))
    w0 = at.lstm.lin.weight.detach().clone()
    at.train()
    assert not torch.equal(w0, at.lstm.lin.weight.detach())
    assert os.path.exists(str(save / "best_lstm.pth.tar")) and os.path.exists(str(save / "valbest_lstm.pth.tar"))
    at.reload_LSTM(str(save / "best_lstm.pth.tar"))
    # extract_late on two synthetic 224x224 frames (hook on features_s, PNG outputs)
    ds = _SPData(2, 224, 4)
    from torch.utils.data import DataLoader
    ds.fs[0] = 1.0
    ds.fs[1] = 0.0                                         # one fixation frame, one saccade frame (LSTM branch)
    at.extract_late(DataLoader(ds, batch_size=1), str(tmp_path / "pred") + "/", str(tmp_path / "feat") + "/")
    assert sorted(os.listdir(str(tmp_path / "pred"))) == ['frame_00000.jpg', 'frame_00001.jpg']
    feat = np.asarray(Image.open(str(tmp_path / "feat" / "frame_00000.jpg")))
    assert feat.shape == (224, 224)
    # LF on files: pred / feat / gt folders with a leave-one-subject-out split by name
    for folder in ("lpred", "lfeat", "lgt"):
        (tmp_path / folder).mkdir()
        rs = np.random.RandomState(len(folder))
        for name in ("Ahmad_a_0001.png", "Ahmad_a_0002.png", "Alireza_a_0001.png"):
            arr = rs.randint(0, 256, (224, 224)).astype(np.uint8)
            if folder == "lgt":
                arr = (synth.synth_gt(1, 224, rs)[0, 0] * 255).astype(np.uint8)
            Image.fromarray(arr).save(str(tmp_path / folder / name))
    lf = LF(save_path=str(save), device='0', late_pred_path=str(tmp_path / "lpred"), num_epoch=1,
            late_feat_path=str(tmp_path / "lfeat"), gt_path=str(tmp_path / "lgt"), val_name='Alireza', batch_size=2,
            lr=1e-4)
    w0 = lf.model.fusion[0].weight.detach().clone()
    lf.train()
    assert not torch.equal(w0, lf.model.fusion[0].weight.detach())
    ck = torch.load(str(save / "best_late.pth.tar"), map_location='cpu', weights_only=False)
    assert set(ck) == {'state_dict', 'loss', 'auc', 'aae'}
    lf.val()
    # glue parity with the golden vectors (AT.py:25-66)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_glue.npz"))
    krs = np.random.RandomState(12)
    krs.standard_normal((64, 3, 3, 3)); [krs.standard_normal((4,)) for _ in range(29)]
    featm = torch.from_numpy(np.abs(krs.standard_normal((2, 512, 14, 14))).astype(np.float32))
    cf = crop_feature(featm, [[5, 220], [117, 60]], 3)
    assert np.array_equal(cf.numpy(), gold["crop_feature"])
    w = cf.contiguous().view(2, 512, -1).mean(2)
    assert np.allclose(get_weighted(w[0], featm[0:1]).numpy(), gold["get_weighted"], rtol=1e-6, atol=1e-7)
