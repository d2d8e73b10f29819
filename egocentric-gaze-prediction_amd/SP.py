"""Mirror of the reference's ``SP.py``: driver of the SP (saliency prediction) module -- weight loading in the three
resume modes, Adam, the train / validation loops and best-val checkpointing (SP.py:18-208).  Same constructor
arguments, attributes and methods (``trainSP``, ``testSP``, ``train``); the model, the loss and the optimizer are the
HIP-backed mirrors (models.model_SP, floss, optim.FusedAdam).  ``device`` is the reference's string index
('0' -> torch.device('cuda:0'), which is the HIP device on PyTorch-ROCm).
"""
import os
import time

import torch
from torch.utils.data import DataLoader

from . import dp
from .floss import BCELoss, floss
from .models.model_SP import model_SP
from .optim import FusedAdam
from .data.STdatas import staged_batches
from .utils import (AverageMeter, cfg, change_key_names, computeAAEAUC, make_layers, owned_state_dict, plot_loss,
                    save_checkpoint)

VGG16_BN_URL = 'https://download.pytorch.org/models/vgg16_bn-6c64b313.pth'


def _progress(it):
    try:
        from tqdm import tqdm
        return tqdm(it)
    except Exception:
        return it


def _load_vgg16_bn():
    """ImageNet VGG16-BN weights (SP.py:54).  Offline: point EGAZE_VGG16_BN at a local copy of the file."""
    local = os.environ.get('EGAZE_VGG16_BN')
    if local:
        return torch.load(local, map_location='cpu', weights_only=False)
    import torch.utils.model_zoo as model_zoo
    return model_zoo.load_url(VGG16_BN_URL)


def _strip_features(d):
    """keep 'features.*' entries and drop the 9-character prefix (SP.py:57-65)."""
    return {k[9:]: v for k, v in d.items() if 'features' in k}


class SP():
    def __init__(self, lr=1e-7, loss_save='loss_SP.png', save_name='best_fusion.pth.tar', save_path='save',
                 loss_function='f', num_epoch=10, batch_size=10, device='0', resume=1,
                 pretrained_spatial='save/04_spatial.pth.tar', pretrained_temporal='save/03_temporal.pth.tar',
                 traindata=None, valdata=None):
        self.lr, self.loss_save, self.save_name, self.save_path = lr, loss_save, save_name, save_path
        os.makedirs(save_path, exist_ok=True)
        self.loss_function, self.num_epoch, self.batch_size = loss_function, num_epoch, batch_size
        self.device = torch.device('cuda:' + device)
        self.pretrained_spatial, self.pretrained_temporal = pretrained_spatial, pretrained_temporal
        # one process per GPU: every rank draws a disjoint 1/world share of each epoch (dp.RankShardSampler); at world 1
        # these are exactly the reference's loaders (SP.py:34-35)
        self.train_sampler = dp.RankShardSampler(traindata, True, batch_size) if dp.world_size() > 1 else None
        val_sampler = dp.RankShardSampler(valdata, False, batch_size, pad=False) if dp.world_size() > 1 else None
        self.STTrainLoader = DataLoader(dataset=traindata, batch_size=batch_size, shuffle=self.train_sampler is None,
                                        sampler=self.train_sampler, num_workers=1, pin_memory=True)
        self.STValLoader = DataLoader(dataset=valdata, batch_size=batch_size, shuffle=False, sampler=val_sampler,
                                      num_workers=1, pin_memory=True)
        in_channels = 20
        self.model = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], in_channels))
        self.epochnow = 0
        pretrained_optimizer = None
        # NB the reference compares against the strings '2' / '0' while its default is the int 1 (SP.py:20,40,51)
        if resume == '2':            # resume from an SP checkpoint, optimizer state included (SP.py:40-50)
            ckpt = torch.load(os.path.join(save_path, save_name), map_location='cpu', weights_only=False)
            self.epochnow = ckpt['epoch']
            pretrained_optimizer = ckpt['optimizer']
            merged = self.model.state_dict()
            merged.update(ckpt['state_dict'])
            self.model.load_state_dict(merged)
        elif resume == '0':          # ImageNet VGG16-BN into both encoders (SP.py:51-73)
            vgg = _load_vgg16_bn()
            flow = _strip_features(change_key_names(vgg, in_channels))
            rgb = _strip_features(vgg)
            sd_s, sd_t = self.model.features_s.state_dict(), self.model.features_t.state_dict()
            sd_s.update({k: v for k, v in rgb.items() if k in sd_s})
            sd_t.update({k: v for k, v in flow.items() if k in sd_t})
            self.model.features_s.load_state_dict(sd_s)
            self.model.features_t.load_state_dict(sd_t)
        else:                        # separately pre-trained streams, encoders frozen (SP.py:74-102)
            ps = torch.load(self.pretrained_spatial, map_location='cpu', weights_only=False)['state_dict']
            pt = torch.load(self.pretrained_temporal, map_location='cpu', weights_only=False)['state_dict']
            ps = {k: v for k, v in ps.items() if 'features' in k}
            pt = {k: v for k, v in pt.items() if 'features' in k}
            sd_s, sd_t = self.model.features_s.state_dict(), self.model.features_t.state_dict()
            # the reference filters with the UN-stripped keys here (SP.py:91-92), so nothing matches and the
            # encoders keep their init before being frozen -- reproduced, not fixed
            sd_s.update({k: v for k, v in ps.items() if k in sd_s})
            sd_t.update({k: v for k, v in pt.items() if k in sd_t})
            self.model.features_s.load_state_dict(sd_s)
            self.model.features_t.load_state_dict(sd_t)
            for p in list(self.model.features_s.parameters()) + list(self.model.features_t.parameters()):
                p.requires_grad = False
        self.model.to(self.device)
        self.criterion = (floss() if loss_function == 'f' else BCELoss()).to(self.device)
        if resume != '0':            # only fusion + bn + decoder are trained (SP.py:109-111)
            params = (list(self.model.fusion.parameters()) + list(self.model.bn.parameters())
                      + list(self.model.decoder.parameters()))
        else:
            params = self.model.parameters()
        self.optimizer = FusedAdam(params, lr=self.lr)
        if pretrained_optimizer is not None:
            self.optimizer.load_state_dict(pretrained_optimizer)
        self.reducer = dp.attach(self.optimizer) if torch.distributed.is_initialized() else None
        from . import hipops
        print(hipops.precision_banner())
        print('SP module init done!')

    def trainSP(self):
        self.model.train()
        batch_time, losses = AverageMeter(), AverageMeter()
        end = time.time()
        self.optimizer.zero_grad()
        # batch k + 1 is copied (and normalised) on a copy stream while step k computes: data.STdatas.staged_batches
        for i, (sample, (input_s, input_t, target)) in _progress(enumerate(staged_batches(self.STTrainLoader, self.device))):
            output = self.model(input_s, input_t)
            loss = self.criterion(output, target.view(output.size()))
            loss.backward()
            self.optimizer.step()
            self.optimizer.zero_grad()
            batch_time.update(time.time() - end)
            losses.update(loss.item(), input_s.size(0))
            end = time.time()
            if (i + 1) % 1000 == 0:
                print('Epoch: [{0}][{1}/{2}]\t''Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                      'Loss {loss.val:.4f} ({loss.avg:.4f})\t'.format(self.epochnow, i + 1, len(self.STTrainLoader) + 1,
                                                                      batch_time=batch_time, loss=losses))
        self.optimizer.check_finite()        # raises if a step of the epoch met NaN / inf gradients (the kernel skipped those elements)
        return losses.avg

    def testSP(self):
        dp.sync_buffers(self.model)                 # all ranks validate rank 0's BN running statistics
        self.model.eval()
        batch_time, losses, auc, aae = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
        end = time.time()
        with torch.no_grad():
            for i, (sample, (input_s, input_t, target)) in _progress(enumerate(staged_batches(self.STValLoader, self.device))):
                output = self.model(input_s, input_t)
                target = target.view(output.size())
                loss = self.criterion(output, target)
                losses.update(loss.item(), input_s.size(0))
                batch_time.update(time.time() - end)
                end = time.time()
                aae1, auc1, _ = computeAAEAUC(output, target)          # device kernel (SP.py:170-174)
                auc.update(auc1)
                aae.update(aae1)
                if (i + 1) % 1000 == 0:
                    print('Test: [{0}/{1}]\t''Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                          'Loss {loss.val:.4f} ({loss.avg:.4f})\t'.format(i, len(self.STValLoader),
                                                                          batch_time=batch_time, loss=losses))
        if dp.world_size() > 1:                      # global averages: every rank takes the same 'best epoch' decision
            loss_avg, auc_avg, aae_avg = dp.reduce_meters((losses.sum, losses.count), (auc.sum, auc.count),
                                                          (aae.sum, aae.count))
        else:
            loss_avg, auc_avg, aae_avg = losses.avg, auc.avg, aae.avg
        if dp.is_main():
            print('AUC: {0}\t AAE: {1}'.format(auc_avg, aae_avg))
        return loss_avg, auc_avg, aae_avg

    def train(self):
        train_loss, val_loss, best_loss = [], [], 100
        for epoch in range(self.epochnow, self.num_epoch):
            self.epochnow = epoch
            if self.train_sampler is not None:
                self.train_sampler.set_epoch(epoch)
            train_loss.append(self.trainSP())
            loss1, auc1, aae1 = self.testSP()
            val_loss.append(loss1)
            if dp.is_main():                         # rank 0 owns the files (SP.py:203-208)
                plot_loss(train_loss, val_loss, os.path.join(self.save_path, self.loss_save))
            if loss1 < best_loss:
                best_loss = loss1
                if dp.is_main():
                    save_checkpoint({'epoch': epoch, 'arch': 'SP', 'state_dict': owned_state_dict(self.model),
                                     'optimizer': self.optimizer.state_dict(), 'auc': auc1, 'aae': aae1},
                                    self.save_name, self.save_path)
            dp.barrier()
