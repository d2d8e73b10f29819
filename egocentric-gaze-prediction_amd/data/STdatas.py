"""Mirror of the reference's ``data/STdatas.py``: RGB frame + 10-pair optical-flow stack + ground-truth gaze map
+ fixation flag per sample (data/STdatas.py:9-73).  Host-side disk I/O, out of the kernel scope; what the hot path
relies on is the tensor contract: 'image' (3,H,W) = (u8/255 - mean)/std on BGR-ordered channels, 'flow' (20,H,W) =
(u8/255 - 0.5)/0.5 ordered x_t, y_t, x_{t-1}, y_{t-1}, ..., 'gt' (1,H,W) = u8/255, 'fixsac' (1,), 'imname'."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from ._io import imread

_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
_STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


def build_temporal_list(imgPath, gtPath, listFolders, listGtFiles):
    """Flow window looks backwards: frames n, n-1, ..., n-9 (data/STdatas.py:18-20); file-name parsing is
    positional like the reference (gt[:-17] = folder, gt[-9:-4] = frame number)."""
    imgx, imgy = [], []
    for gt in listGtFiles:
        folder, number = gt[:-17], int(gt[-9:-4])
        assert folder in listFolders
        imgx.append([os.path.join(imgPath, folder, 'flow_x_%05d.jpg' % (number - m)) for m in range(10)])
        imgy.append([os.path.join(imgPath, folder, 'flow_y_%05d.jpg' % (number - m)) for m in range(10)])
    return imgx, imgy


IMAGE_MEAN, IMAGE_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # applied to BGR-ordered channels (:51-55)
FLOW_MEAN, FLOW_STD = (0.5,) * 20, (0.5,) * 20                             # (:59-68)


def stage_batch(sample, device):
    """('image', 'flow', 'gt') of a collated sample as normalised fp32 tensors on ``device``.  A dataset built with
    ``raw_u8=True`` hands over bytes: they cross PCIe at a quarter of the fp32 size and are normalised by
    ``egz_u8_normalize`` with the reference's own three fp32 operations (bit-exact with the host expression below)."""
    if sample['image'].dtype == torch.uint8:
        from .. import hipops as H
        put = lambda t: t.contiguous().to(device, non_blocking=True)
        return (H.u8_normalize(put(sample['image']), IMAGE_MEAN, IMAGE_STD),
                H.u8_normalize(put(sample['flow']), FLOW_MEAN, FLOW_STD),
                H.u8_normalize(put(sample['gt']), (0.0,), (1.0,)))
    return (sample['image'].float().to(device, non_blocking=True), sample['flow'].float().to(device, non_blocking=True),
            sample['gt'].float().to(device, non_blocking=True))


def staged_batches(loader, device, stage=None):
    """Iterates ``loader`` one batch ahead: yields ``(sample, (image, flow, gt))`` with the NEXT batch's host-to-device copy
    and normalisation (stage_batch) already issued on a copy stream, so that batch k + 1 crosses PCIe while step k computes
    (the reference copies inside the step, SP.py:126-131).  Ordering is by stream events only; the yielded tensors are
    handed to the consumer's stream (record_stream) so the caching allocator cannot recycle them early.  Falls back to
    in-step staging when HIP streams are switched off (EGAZE_STREAMS=0) or the device is not a GPU.  ``stage(sample,
    device) -> tuple of device tensors`` defaults to stage_batch (the SP / AT sample layout); LF passes its own."""
    from .. import streams
    stage = stage or stage_batch
    device = torch.device(device)
    if device.type != 'cuda' or not streams.ENABLED:
        for sample in loader:
            yield sample, stage(sample, device)
        return
    copy = streams.side_stream("h2d")

    def issue(sample):
        with torch.cuda.stream(copy):
            staged = stage(sample, device)
            if stage is stage_batch:
                # the flow stack's NHWC-32 form and its abs-max, behind the copy on the same stream (hipops.prepare_network_input):
                # two weight-independent HBM passes less at the head of the consumer's forward pass
                from .. import hipops
                hipops.prepare_network_input(staged[1])
        ev = torch.cuda.Event()
        ev.record(copy)
        return sample, staged, ev

    it = iter(loader)
    try:
        nxt = issue(next(it))
    except StopIteration:
        return
    while nxt is not None:
        sample, staged, ev = nxt
        try:
            nxt = issue(next(it))          # issued BEFORE the consumer's kernels of this batch: overlaps with them
        except StopIteration:
            nxt = None
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in staged:
            t.record_stream(cur)
        yield sample, staged


class STDataset(Dataset):
    def __init__(self, imgPath, imgPath_s, gtPath, listFolders, listTrainFiles, listGtFiles, listfixsacTrain,
                 fixsacPath, raw_u8=False):
        self.raw_u8 = raw_u8
        self.listFolders, self.listGtFiles = listFolders, listGtFiles
        self.imgPath, self.imgPath_s, self.gtPath = imgPath, imgPath_s, gtPath
        self.listTrainFiles = listTrainFiles
        self.imgx, self.imgy = build_temporal_list(imgPath, gtPath, listFolders, listGtFiles)
        chunks = []
        for f in listfixsacTrain:          # dilate the fixation labels by one frame each side (:37-41)
            a = np.loadtxt(os.path.join(fixsacPath, f))
            chunks.append((np.convolve(a, np.array([1, 1, 1]))[1:-1] > 0).astype(float))
        self.fixsac = np.concatenate(chunks) if chunks else np.zeros(0)

    def __len__(self):
        return len(self.listGtFiles)

    def __getitem__(self, index):
        im = torch.from_numpy(imread(os.path.join(self.imgPath_s, self.listTrainFiles[index])).transpose((2, 0, 1)).copy())
        if self.raw_u8:            # bytes out; stage_batch() normalises on the device
            planes = []
            for fx, fy in zip(self.imgx[index], self.imgy[index]):
                planes.append(torch.from_numpy(imread(fx, gray=True)))
                planes.append(torch.from_numpy(imread(fy, gray=True)))
            gt = torch.from_numpy(imread(os.path.join(self.gtPath, self.listGtFiles[index]), gray=True))
            return {'image': im, 'flow': torch.stack(planes), 'gt': gt.unsqueeze(0),
                    'fixsac': torch.FloatTensor([self.fixsac[index]]), 'imname': self.listTrainFiles[index]}
        im = (im.float().div(255) - _MEAN) / _STD
        planes = []
        for fx, fy in zip(self.imgx[index], self.imgy[index]):
            planes.append(torch.from_numpy(imread(fx, gray=True)))
            planes.append(torch.from_numpy(imread(fy, gray=True)))
        flow = (torch.stack(planes).float().div_(255) - 0.5) / 0.5
        gt = torch.from_numpy(imread(os.path.join(self.gtPath, self.listGtFiles[index]), gray=True)).float().div(255)
        return {'image': im, 'flow': flow, 'gt': gt.unsqueeze(0),
                'fixsac': torch.FloatTensor([self.fixsac[index]]), 'imname': self.listTrainFiles[index]}
