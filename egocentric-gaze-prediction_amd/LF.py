"""Mirror of the reference's ``LF.py``: driver of the late-fusion module (LF.py:16-161) -- file listing with
leave-one-subject-out split, Adam, train / validation loops with per-batch AAE / AUC, best-train and best-val
checkpoints.  Model, loss and optimizer are the HIP-backed mirrors."""
import os

import torch
from torch.utils.data import DataLoader

from . import dp
from .data.lateDataset import lateDataset
from .floss import BCELoss, floss
from .models.late_fusion import late_fusion
from .optim import FusedAdam
from .utils import AverageMeter, aae_auc_from_rows, aae_auc_rows, computeAAEAUC, owned_state_dict, plot_loss
from .SP import _progress

LF_GRAPH = os.environ.get("EGAZE_LF_GRAPH", "1") != "0"      # A/B knob: 0 = issue every LF training step launch by launch


def _split(folder, val_name, task):
    names = os.listdir(folder)
    train = sorted(k for k in names if val_name not in k)
    val = sorted(k for k in names if val_name in k)
    if task is not None:
        train = [k for k in train if task in k]
        val = [k for k in val if task in k]
    return train, val


class GraphedLateIteration:
    """LF.trainLate's iteration (LF.py:90-100) -- late_fusion forward, loss, the batch's AAE / AUC (computeAAEAUC, LF.py:92-94),
    zero_grad, backward, Adam -- as ONE hipGraph replay (graphs.GraphedTrainStep), the metric kernel included.  The reference
    reads the loss and both maps back every iteration only to feed three running averages; here the loss and the metric's 6
    doubles per sample are copied into pinned host slots asynchronously and read RING iterations at a time (``drain``): the
    same values reach the same meters in the same order, but the host no longer waits for the device after every replay
    (the per-iteration read-back cost 0.22 ms of a 1.44 ms iteration, profiles/r04_lf_step.txt)."""
    RING = 16

    def __init__(self, model, criterion, optimizer, example):
        from .graphs import GraphedTrainStep

        def fwd_loss(feat_, im_, gt_):
            o = model(feat_, im_)                             # channel 0 = AT map, channel 1 = SP map (LF.py:90)
            l = criterion(o, gt_)
            rows, _ = aae_auc_rows(o, gt_)                    # device kernel, maps stay in HBM (LF.py:92-94)
            return l, o, rows
        self.step = GraphedTrainStep(fwd_loss, optimizer, example)
        self.shape = tuple(example[0].shape)
        self.h_rows = torch.empty((self.RING, self.shape[0], 6), dtype=torch.float64).pin_memory()
        self.h_loss = torch.empty((self.RING,), dtype=torch.float32).pin_memory()
        self.pending = 0

    def __call__(self, feat, im, gt):
        loss, out, rows = self.step(feat, im, gt)
        self.h_rows[self.pending].copy_(rows, non_blocking=True)
        self.h_loss[self.pending].copy_(loss, non_blocking=True)
        self.pending += 1
        return out

    @property
    def full(self):
        return self.pending >= self.RING

    def drain(self):
        """-> [(loss, aae, auc)] of the iterations since the last drain, oldest first (one synchronisation for all of them)."""
        if not self.pending:
            return []
        torch.cuda.current_stream().synchronize()
        done = []
        for k in range(self.pending):
            aae1, auc1, _ = aae_auc_from_rows(self.h_rows[k].numpy())
            done.append((float(self.h_loss[k]), aae1, auc1))
        self.pending = 0
        return done

    def close(self):
        self.step.close()


def _stage_late(sample, device):
    return tuple(sample[k].float().to(device, non_blocking=True) for k in ('im', 'gt', 'feat'))


class LF():
    def __init__(self, pretrained_model=None, save_path='save', late_save_img='loss_late.png',
                 save_name='best_late.pth.tar', device='0', late_pred_path='../new_pred', num_epoch=10,
                 late_feat_path='../new_feat', gt_path='../gtea_gts', val_name='Alireza', batch_size=32,
                 loss_function='f', lr=1e-7, task=None):
        self.model = late_fusion()
        self.device = torch.device('cuda:' + device)
        self.save_name, self.save_path = save_name, save_path
        if pretrained_model is not None:
            merged = self.model.state_dict()
            merged.update(torch.load(pretrained_model, map_location='cpu', weights_only=False)['state_dict'])
            self.model.load_state_dict(merged)
            print('loaded pretrained late fusion model from ' + pretrained_model)
        self.model.to(self.device)
        self.batch_size, self.num_epoch, self.epochnow, self.late_save_img = batch_size, num_epoch, 0, late_save_img
        listGtFiles, listValGtFiles = _split(gt_path, val_name, task)
        print('num of training LF samples: %d' % len(listGtFiles))
        print('Loading SP predictions from /%s' % late_pred_path)
        listTrainFiles, listValFiles = _split(late_pred_path, val_name, task)
        print('num of LF val samples: ', len(listValFiles))
        listTrainFeats, listValFeats = _split(late_feat_path, val_name, task)
        assert(len(listTrainFeats) == len(listTrainFiles) and len(listGtFiles) > 0)
        assert(len(listValGtFiles) == len(listValFiles))
        train_set = lateDataset(late_pred_path, gt_path, late_feat_path, listTrainFiles, listGtFiles, listTrainFeats)
        val_set = lateDataset(late_pred_path, gt_path, late_feat_path, listValFiles, listValGtFiles, listValFeats)
        # rank-sharded under torch.distributed (dp.RankShardSampler); the reference's loaders (LF.py:69-72) at world 1
        self.train_sampler = dp.RankShardSampler(train_set, True, batch_size) if dp.world_size() > 1 else None
        val_sampler = dp.RankShardSampler(val_set, False, batch_size, pad=False) if dp.world_size() > 1 else None
        self.train_loader = DataLoader(dataset=train_set, batch_size=batch_size, shuffle=self.train_sampler is None,
                                       sampler=self.train_sampler, num_workers=0, pin_memory=True)
        self.val_loader = DataLoader(dataset=val_set, batch_size=batch_size, shuffle=False, sampler=val_sampler,
                                     num_workers=0, pin_memory=True)
        self.criterion = (floss() if loss_function == 'f' else BCELoss()).to(self.device)
        self.optimizer = FusedAdam(self.model.parameters(), lr=lr)
        self.reducer = dp.attach(self.optimizer) if torch.distributed.is_initialized() else None
        from . import hipops
        print(hipops.precision_banner())

    def _run(self, loader, train, every):
        from .data.STdatas import staged_batches
        losses, auc, aae = AverageMeter(), AverageMeter(), AverageMeter()
        # the three maps of batch k + 1 cross PCIe on a copy stream while step k computes (LF.py:85-89 copies in the step)
        # The training step (forward + loss + zero_grad + backward + Adam, ~110 launches for ~1.4 ms of kernels) is captured
        # once and replayed (graphs.GraphedTrainStep); single process only -- the gradient reducer's hooks are host code.
        graphed = None
        use_graph = train and LF_GRAPH and dp.world_size() == 1 and self.device.type == 'cuda'
        try:
            return self._run_loop(loader, train, every, use_graph, losses, auc, aae)
        finally:
            # an exception / KeyboardInterrupt inside the loop must not leave the optimizer capturable with a stale host step
            # count: close() syncs the count back from the device and leaves capturable mode
            g = getattr(self, "_graphed", None)
            if g is not None:
                g.close()
                self._graphed = None

    def _run_loop(self, loader, train, every, use_graph, losses, auc, aae):
        from .data.STdatas import staged_batches
        graphed = None

        def _update(done):
            for loss1, aae1, auc1 in done:
                auc.update(auc1)
                aae.update(aae1)
                losses.update(loss1)
        for i, (sample, (im, gt, feat)) in _progress(enumerate(staged_batches(loader, self.device, _stage_late))):
            if use_graph and graphed is None:
                graphed = self._graphed = GraphedLateIteration(self.model, self.criterion, self.optimizer, (feat, im, gt))
            if graphed is not None and tuple(feat.shape) == graphed.shape:
                graphed(feat, im, gt)                         # one replay = one LF.trainLate iteration (LF.py:90-100)
                if graphed.full or (i + 1) % every == 0:
                    _update(graphed.drain())
                if (i + 1) % every == 0:
                    print('Epoch: [{0}][{1}/{2}]\t''AUCAAE_late {auc.avg:.3f} ({aae.avg:.3f})\t'
                          'Loss {loss.val:.4f} ({loss.avg:.4f})\t'.format(self.epochnow, i + 1, len(loader) + 1, auc=auc,
                                                                          loss=losses, aae=aae))
                continue
            if graphed is not None:
                _update(graphed.drain())                      # (a trailing partial batch: the meters stay in iteration order)
            out = self.model(feat, im)                       # channel 0 = AT map, channel 1 = SP map (LF.py:90)
            loss = self.criterion(out, gt)
            aae1, auc1, _ = computeAAEAUC(out.detach(), gt)          # device kernel, maps stay in HBM (LF.py:92-94)
            auc.update(auc1)
            aae.update(aae1)
            losses.update(loss.item())
            if train:
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()
            if (i + 1) % every == 0:
                print('Epoch: [{0}][{1}/{2}]\t''AUCAAE_late {auc.avg:.3f} ({aae.avg:.3f})\t'
                      'Loss {loss.val:.4f} ({loss.avg:.4f})\t'.format(self.epochnow, i + 1, len(loader) + 1, auc=auc,
                                                                      loss=losses, aae=aae))
        if graphed is not None:
            _update(graphed.drain())
        if train:
            self.optimizer.check_finite()            # (NaN / inf gradient elements are skipped by the Adam kernel and reported here)
        if dp.world_size() > 1:                      # global averages so that every rank agrees on the best epoch
            return tuple(dp.reduce_meters((losses.sum, losses.count), (auc.sum, auc.count), (aae.sum, aae.count)))
        return losses.avg, auc.avg, aae.avg

    def trainLate(self):
        # like the reference, the model is never switched to eval(): BN uses batch statistics in both loops
        return self._run(self.train_loader, True, 3000)

    def testLate(self):
        dp.sync_buffers(self.model)                  # train-mode BN here too (never eval()), but checkpoints are rank 0's
        with torch.no_grad():
            return self._run(self.val_loader, False, 1000)

    def train(self):
        print('begin training LF module...')
        trainprev, valprev, loss_train, loss_val = 999, 999, [], []
        for epoch in range(self.num_epoch):
            self.epochnow = epoch
            if self.train_sampler is not None:
                self.train_sampler.set_epoch(epoch)
            loss, auc, aae = self.trainLate()
            loss_train.append(loss)
            print('training, auc is %5f, aae is %5f' % (auc, aae))
            if loss < trainprev:
                if dp.is_main():                      # rank 0 owns the files (LF.py:144-153)
                    torch.save({'state_dict': owned_state_dict(self.model), 'loss': loss, 'auc': auc, 'aae': aae},
                               os.path.join(self.save_path, self.save_name))
                trainprev = loss
            loss, auc, aae = self.testLate()
            loss_val.append(loss)
            if dp.is_main():
                plot_loss(loss_train, loss_val, os.path.join(self.save_path, self.late_save_img))
            if loss < valprev:
                if dp.is_main():
                    torch.save({'state_dict': owned_state_dict(self.model), 'loss': loss, 'auc': auc, 'aae': aae},
                               os.path.join(self.save_path, 'val' + self.save_name))
                valprev = loss
            dp.barrier()
            print('testing, auc is %5f, aae is %5f' % (auc, aae))
        print('LF module training finished!')

    def val(self):
        print('begin testing LF module...')
        loss, auc, aae = self.testLate()
        print('AUC is : %04f, AAE is: %04f' % (auc, aae))
        print('LF module testing finished!')
