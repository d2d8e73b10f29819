"""Mirror of the reference's ``extractLSTMw.py`` (offline data preparation between SP and AT, extractLSTMw.py:21-138):
run ``features_s`` alone over a dataset and store, for the second frame of every fixation, the spatial mean of a window
of the conv5_3 map around the ground-truth gaze cell as ``fix_<name>.pth.tar``.  The encoder forward is the same fused
HIP path the SP model uses; the window mean is one small kernel (egz_window_mean).

Reproduced, not "fixed" (checked against the reference by tests/golden/extract_lstm.npz):
  * the gaze cell is the arg-max of the ``AvgPool2d(16)``-downsampled ground truth (extractLSTMw.py:66,85-88), not of the
    full-resolution map;
  * ``crop_feature_var`` clips with FLOAT bounds and slices with ``int()`` (extractLSTMw.py:46-58): for crop_size 3 the
    window is rows/cols [int(f - 1.5), int(f + 2)) with f clipped to [1.5, H - 2] -- 4 x 4 cells in the interior and
    3 at the low border, unlike AT.crop_feature's 3 x 3;
  * the fixation state machine (extractLSTMw.py:74-111): the first frame of a fixation arms it, the second is extracted,
    the rest are skipped until a saccade frame; a fixation that ends after one frame raises RuntimeError.
"""
import math
import os

import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from .utils import cfg, make_layers


def var_window(ind, size, H, W):
    """Half-open cell window (y0, y1, x0, x1) of extractLSTMw.crop_feature_var for the flat arg-max ``ind`` of an
    H x W map: float clip, then int() truncation of both slice bounds (extractLSTMw.py:52-54)."""
    up = int(math.ceil(size / 2.0))
    out = []
    for f, n in ((ind // W, H), (ind % W, H)):          # the reference clips both coordinates with H (square maps)
        f = min(max(float(f), size / 2), float(n - up))
        out += [int(f - size / 2), int(f + up)]
    return tuple(out)


def crop_feature_var(feature, maxind, size):
    """(B,C,H,W) map, (B,...) flat arg-max indices -> the windows stacked along the batch dim (all samples of a batch
    must produce the same window shape, as in the reference's torch.cat)."""
    H, W = feature.size(2), feature.size(3)
    out = []
    for b in range(feature.size(0)):
        y0, y1, x0, x1 = var_window(int(maxind[b].item()), size, H, W)
        out.append(feature[b:b + 1, :, y0:y1, x0:x1])
    return torch.cat(out, 0)


def crop_feature_align(feature, maxind, size):
    """``--align`` variant on the x16 bilinearly upsampled map (extractLSTMw.py:32-44): integer clip, size*16 window."""
    size *= 16
    H, W = feature.size(2), feature.size(3)
    out = []
    for b in range(feature.size(0)):
        ind = int(maxind[b].item())
        fy = min(max(ind // W, size // 2), H - size // 2)
        fx = min(max(ind % W, size // 2), H - size // 2)
        out.append(feature[b:b + 1, :, fy - size // 2:fy + size // 2, fx - size // 2:fx + size // 2])
    return torch.cat(out, 0)


def channel_weight(feat, gt, crop_size, align):
    """chn_weight (512,) of one sample: feat (1,512,14,14) on the device, gt (1,1,224,224) on the host."""
    if align:
        flat = gt.float().view(gt.size(0), gt.size(1), -1)
        _, maxind = torch.max(flat, 2)
        if feat.is_cuda:             # the window mean of the x16 bilinear upsampling as one kernel on the 14 x 14 map
            from .AT import crop_align_mean
            full = feat.size(3) * 16
            gp = [[int(maxind[b].item()) // full, int(maxind[b].item()) % full] for b in range(feat.size(0))]
            return crop_align_mean(feat, gp, crop_size).squeeze(0)
        up = nn.functional.interpolate(feat.contiguous(), scale_factor=16, mode='bilinear', align_corners=True)
        crop = crop_feature_align(up, maxind, crop_size).contiguous()
        return crop.view(crop.size(0), crop.size(1), -1).mean(2).squeeze(0)
    pooled = nn.functional.avg_pool2d(gt.float(), 16)                  # the reference's nn.AvgPool2d(16) (host: 14 x 14)
    _, maxind = torch.max(pooled.view(pooled.size(0), pooled.size(1), -1), 2)
    win = [var_window(int(maxind[b].item()), crop_size, feat.size(2), feat.size(3)) for b in range(feat.size(0))]
    if feat.is_cuda:
        from . import hipops as H
        from .functions import to_nhwc
        return H.window_mean(to_nhwc(feat), win).squeeze(0)
    crop = crop_feature_var(feat, maxind, crop_size).contiguous()
    return crop.view(crop.size(0), crop.size(1), -1).mean(2).squeeze(0)


def extractw(loader, model, savepath, crop_size=3, device='0', align=False):
    dev = torch.device(device if str(device).startswith(('cuda', 'cpu')) else 'cuda:' + str(device))
    print('extracting lstm training data...')
    os.makedirs(savepath, exist_ok=True)
    OUT, ARMED, TAKEN = 0, 1, 2           # no fixation / first fixation frame seen / second frame extracted
    state = OUT
    with torch.no_grad():
        for i, sample in enumerate(loader):
            fix = float(sample['fixsac']) == 1.0
            if state == OUT:
                state = ARMED if fix else OUT
            elif state == ARMED:
                if not fix:
                    raise RuntimeError('fixation is not processed.')       # extractLSTMw.py:110-111
                state = TAKEN
                feat = model(sample['image'].float().to(dev))                      # (1,512,14,14)
                w = channel_weight(feat, sample['gt'], crop_size, align).cpu()
                torch.save(w, os.path.join(savepath, 'fix_' + sample['imname'][0][:-4] + '.pth.tar'))
            elif not fix:
                state = OUT
    print('done')


def extract_LSTM_training_data(save_path='../512w', trained_model='save/best_fusion.pth.tar', device='0', crop_size=3,
                               traindata=None, valdata=None, align=False):
    model = make_layers(cfg['D'], 3)
    sd = torch.load(trained_model, map_location='cpu', weights_only=False)['state_dict']
    own = model.state_dict()
    own.update({k[len('features_s.'):]: v for k, v in sd.items() if k.startswith('features_s.')})
    model.load_state_dict(own)
    model.to(torch.device('cuda:' + device)).eval()
    for data, sub in ((traindata, 'train'), (valdata, 'test')):
        loader = DataLoader(dataset=data, batch_size=1, shuffle=False, num_workers=1, pin_memory=True)
        extractw(loader, model, os.path.join(save_path, sub), crop_size, device, align)
    print('Attention weight for training LSTMnet successfully extracted.')
