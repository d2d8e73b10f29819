"""Mirror of the reference's ``extractLSTMw.py`` (offline data preparation between SP and AT, extractLSTMw.py:21-138):
run ``features_s`` alone over a dataset and store, for the second frame of every fixation, the spatial mean of the
crop_size x crop_size window of the conv5_3 map around the ground-truth gaze point as ``fix_<name>.pth.tar``.
The encoder forward is the same fused HIP path the SP model uses."""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from .utils import cfg, make_layers


def crop_feature_var(feature, maxind, size):
    from .AT import crop_feature
    return crop_feature(feature, maxind, size)


def crop_feature_align(feature, maxind, size):
    from .AT import crop_align_feature
    return crop_align_feature(feature, maxind, size)


def extractw(loader, model, savepath, crop_size=3, device='0', align=False):
    dev = torch.device('cuda:' + device)
    os.makedirs(savepath, exist_ok=True)
    prev_fix = 0
    run = 0
    with torch.no_grad():
        for i, sample in enumerate(loader):
            fixsac = int(sample['fixsac'])
            run = run + 1 if fixsac == 1 else 0
            if run != 2:                         # the second frame of each fixation
                continue
            feat = model(sample['image'].float().to(dev))                      # (1,512,14,14)
            gt = sample['gt'].numpy().squeeze()
            gp = [list(np.unravel_index(gt.argmax(), gt.shape))]
            crop = (crop_feature_align if align else crop_feature_var)(feat, gp, crop_size).contiguous()
            w = crop.view(crop.size(0), crop.size(1), -1).mean(2).squeeze(0).cpu()
            torch.save(w, os.path.join(savepath, 'fix_' + sample['imname'][0][:-4] + '.pth.tar'))


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
