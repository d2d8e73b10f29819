"""Host-side mirror of the reference's ``utils.py`` (same names, arguments and behaviour).

``make_layers`` / ``cfg`` build the VGG16-BN encoder with the reference's child indices and
state-dict keys (utils.py:57-76), but the returned ``nn.Sequential`` subclass executes fused
[conv3x3 -> BatchNorm -> ReLU (-> max-pool)] blocks on the HIP kernels instead of calling its
children one by one.  Host glue (AverageMeter, repackage_hidden, change_key_names, computeAAEAUC,
plot_loss, save_checkpoint, generalException) keeps the reference semantics.
"""
from __future__ import annotations

import collections
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import hipops as H
from .functions import ConvBNReLUPool, ConvReLU, HeadSigmoid

# utils.py:57-62 -- last max pooling removed
cfg = {
    'A': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512],
    'B': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512],
    'D': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512],
    'E': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512],
}


class generalException(Exception):
    pass


def _bn_args(bn: nn.BatchNorm2d):
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not used by the reference")
    return bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, float(bn.momentum), float(bn.eps)


def conv_bn_relu_pool(x, conv: nn.Conv2d, bn: nn.BatchNorm2d, pool: bool, first: bool, out_buf=None, next_k: int = 0):
    """One fused encoder block on the HIP path; keeps BatchNorm's num_batches_tracked bookkeeping.
    ``next_k``: filters of the conv-BN-ReLU block that consumes the output (0 = none): lets a narrow block leave its
    [BN -> ReLU] to that block (functions.ConvBNReLUPool)."""
    if conv.kernel_size != (3, 3) or conv.padding != (1, 1) or conv.stride != (1, 1):
        raise NotImplementedError("only 3x3 / pad 1 / stride 1 convolutions are on the reference path")
    g, b, rm, rv, training, mom, eps = _bn_args(bn)
    # num_batches_tracked += 1 happens inside the statistics kernel (egz_bn_finalize), not as a separate launch
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
    H.INFER_CALL = not torch.is_grad_enabled()     # no graph will be built: inference (eval blocks may fold their BatchNorm)
    return ConvBNReLUPool.apply(x, conv.weight, conv.bias, g, b, rm, rv, training, mom, eps, pool, first, out_buf, nbt,
                                next_k)


class FusedSequential(nn.Sequential):
    """nn.Sequential holding exactly the reference's children (so child indices, state-dict keys, hooks and
    ``.parameters()`` are identical) whose ``forward`` executes fused HIP blocks instead of the children:

      Conv2d 3x3 -> BatchNorm2d -> ReLU [-> MaxPool2d(2,2)]   -> functions.ConvBNReLUPool
      [Upsample x2 nearest ->] Conv2d 3x3 -> ReLU             -> functions.ConvReLU (upsample folded in)
      Conv2d 1x1 (C -> 1) as last child, followed by Sigmoid  -> functions.HeadSigmoid (``fuse_sigmoid=True``)

    Used for the VGG16-BN encoders (utils.py:64-76), the SP decoder (models/model_SP.py:13-31) and the
    late-fusion stack (models/late_fusion.py:10-13).  A first conv with < 32 input channels reads the NCHW
    network input directly; every later tensor is channels_last."""

    def forward(self, x, fuse_sigmoid=False, out_buf=None, after_first_block=None):
        """``out_buf``: optional NHWC destination for the output of the LAST block when that block is a conv-BN-ReLU
        block (model_SP passes the two encoders the halves of one buffer, see functions.FusionBlock).
        ``after_first_block``: optional callable run once the first block's kernels have been issued (model_SP records a
        stream event there when its encoder stagger is on)."""
        for x in self.blocks(x, fuse_sigmoid, out_buf, after_first_block):
            pass
        return x

    def blocks(self, x, fuse_sigmoid=False, out_buf=None, after_first_block=None):
        """Generator form of ``forward``: issues one fused block per ``next()`` and yields its output (the last value is the
        stack's output).  model_SP drives its two encoders alternately with it, each on its own HIP stream, so that the host
        feeds both streams at the same pace instead of issuing one encoder's ~100 launches before the other's first."""
        mods = list(self.children())
        i, n = 0, len(mods)
        first, ups = True, False
        relu_below = False           # the current tensor is the output of a ConvReLU block of THIS stack
        while i < n:
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < n else None
            if isinstance(m, nn.Upsample):
                if m.scale_factor not in (2, 2.0) or m.mode != 'nearest':
                    raise NotImplementedError("only nearest x2 upsampling is on the reference path")
                ups = True
                i += 1
                continue
            if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and isinstance(nxt, nn.BatchNorm2d) \
                    and i + 2 < n and isinstance(mods[i + 2], nn.ReLU):
                if ups:
                    raise NotImplementedError("upsample in front of a BatchNorm block is not on the reference path")
                pool = i + 3 < n and isinstance(mods[i + 3], nn.MaxPool2d)
                step = 4 if pool else 3
                # the consumer of this block's output, when it is another conv-BN-ReLU block of this stack
                nm = mods[i + step] if i + step + 2 < n else None
                next_k = nm.out_channels if (isinstance(nm, nn.Conv2d) and nm.kernel_size == (3, 3) and nm.padding == (1, 1)
                                             and nm.stride == (1, 1) and nm.in_channels == m.out_channels
                                             and isinstance(mods[i + step + 1], nn.BatchNorm2d)
                                             and isinstance(mods[i + step + 2], nn.ReLU)
                                             # (a consumer whose BatchNorm is frozen while this one trains -- mixed-mode
                                             # fine-tuning -- cannot take deferred / pre-split input: it gets plain fp32)
                                             and mods[i + step + 1].training == nxt.training) else 0
                x = conv_bn_relu_pool(x, m, nxt, pool, first and m.in_channels < 32,
                                      out_buf if i + step >= n else None, next_k)
                relu_below = False
                i += step
            elif isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and isinstance(nxt, nn.ReLU):
                x = ConvReLU.apply(x, m.weight, m.bias, ups, relu_below)
                ups = False
                relu_below = True
                i += 2
            elif isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.out_channels == 1 and i == n - 1:
                if not fuse_sigmoid:
                    raise NotImplementedError("the 1x1 head runs fused with the Sigmoid that follows it")
                x = HeadSigmoid.apply(x, m.weight, m.bias, relu_below)
                i += 1
            else:
                raise NotImplementedError(f"layer pattern at index {i} ({type(m).__name__}) is not on the HIP path")
            if first and after_first_block is not None:
                after_first_block()
            first = False
            yield x



def make_layers(cfg, in_channels, batch_norm=True):
    """utils.make_layers (utils.py:64-76): same children, same indices, fused execution."""
    layers = []
    for v in cfg:
        if v == 'M':
            layers += [nn.MaxPool2d(kernel_size=2, stride=2)]
        else:
            conv2d = nn.Conv2d(in_channels, v, kernel_size=3, padding=1)
            if batch_norm:
                layers += [conv2d, nn.BatchNorm2d(v), nn.ReLU(inplace=False)]
            else:
                layers += [conv2d, nn.ReLU(inplace=True)]
            in_channels = v
    return FusedSequential(*layers)


def init_like_reference(root: nn.Module):
    """The init both model_SP and late_fusion apply to every sub-module (models/model_SP.py:52-65,
    models/late_fusion.py:25-38): Conv2d ~ N(0, sqrt(2/(kh*kw*Cout))) (fan-out), zero bias; BatchNorm
    gamma 1 / beta 0; Linear ~ N(0, 0.01).  Conv3d is not matched and keeps torch's default init."""
    for m in root.modules():
        if isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2. / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0, 0.01)
            m.bias.data.zero_()


def owned_state_dict(module):
    """``module.state_dict()`` with every tensor cloned: the optimizer re-homes all parameters into ONE flat buffer
    (optim.FusedAdam), and torch.save of views would write that shared storage and make every parameter of a re-loaded
    checkpoint alias one buffer.  Checkpoints written by the drivers are storage-independent like the reference's."""
    return collections.OrderedDict((k, v.detach().clone()) for k, v in module.state_dict().items())


def save_checkpoint(state, filename, save_path):
    torch.save(state, os.path.join(save_path, filename))


def var_to_image(var):
    ten = var.data.cpu()
    if ten.dim() == 4:
        ten = ten[0, :, :, :].squeeze()
    if ten.dim() == 3:
        ten = ten.mul(torch.FloatTensor([0.229, 0.224, 0.225]).view(3, 1, 1))
        ten = ten.add(torch.FloatTensor([0.485, 0.456, 0.406]).view(3, 1, 1))
        return ten.numpy().transpose((1, 2, 0))
    elif ten.dim() == 2:
        return ten.numpy()
    print('warning: input variable is invalid to transfer to image')
    return np.zeros((224, 224))


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def repackage_hidden(h):
    """Detach the LSTM state from its history (utils.py:49-55)."""
    if isinstance(h, tuple):
        return tuple(repackage_hidden(v) for v in h)
    if h is None:
        return None
    return h.detach()


def change_key_names(old_params, in_channels):
    """utils.change_key_names (utils.py:78-94): first 25 entries of a vgg16_bn state dict; entry 0
    (RGB conv1 weight) -> mean over RGB repeated ``in_channels`` times for the flow stream."""
    new_params = collections.OrderedDict()
    for n, (key, val) in enumerate(old_params.items()):
        if n >= 25:
            break
        if n == 0:
            new_params[key] = torch.mean(val, dim=1, keepdim=True).repeat(1, in_channels, 1, 1)
        else:
            new_params[key] = val
    return new_params


def _aae_auc_one(out_sq, tar_sq, npix):
    from scipy import ndimage
    predicted = ndimage.center_of_mass(out_sq)
    (i, j) = np.unravel_index(tar_sq.argmax(), tar_sq.shape)
    d = 112 / math.tan(math.pi / 6)
    r1 = np.array([predicted[0] - 112, predicted[1] - 112, d])
    r2 = np.array([i - 112, j - 112, d])
    angle = math.degrees(math.atan2(np.linalg.norm(np.cross(r1, r2)), np.dot(r1, r2)))
    z = np.zeros((224, 224))
    z[int(predicted[0])][int(predicted[1])] = 1
    z = ndimage.gaussian_filter(z, 14)
    z = z - np.min(z)
    z = z / np.max(z)
    fp = (z > z[i][j]).sum()
    return angle, 1 - float(fp) / npix, [i, j]


def aae_auc_rows(output, target):
    """The device half of computeAAEAUC: GPU maps in, a (B, 6) float64 DEVICE tensor out (csrc/metrics.hip: AAE deg, fp count,
    gaze row / col, centroid row / col per sample) -- no read-back, capturable into a hipGraph (LF._run parks the rows of
    several iterations and reads them back together).  Returns (rows, single)."""
    from . import hipops as H
    o = output.detach().to(torch.float32).squeeze()
    t = target.detach().to(torch.float32).squeeze()
    single = o.ndim == 2
    if single:
        o, t = o.unsqueeze(0), t.unsqueeze(0)
    return H.aae_auc(o.contiguous(), t.contiguous()), single


def aae_auc_from_rows(res, single=False, npix_h=224, npix_w=224):
    """The host half: the rows of aae_auc_rows (a numpy (B, 6) array) -> computeAAEAUC's return values."""
    gp = [[int(r[2]), int(r[3])] for r in res]
    if single:
        return float(res[0, 0]), 1 - float(res[0, 1]) / (npix_h * npix_w), gp
    aae = [float(r[0]) for r in res]
    auc = [1 - float(r[1]) / npix_w / npix_h for r in res]
    return np.mean(aae), np.mean(auc), gp


def _aae_auc_device(output, target):
    """GPU tensors in, same return values: the maps stay in HBM, one kernel per batch (csrc/metrics.hip), 6 doubles
    per sample come back.  Shapes as the reference's callers produce them after ``.squeeze()``: (B,224,224) or (224,224)."""
    rows, single = aae_auc_rows(output, target)
    return aae_auc_from_rows(rows.cpu().numpy(), single, int(output.shape[-2]), int(output.shape[-1]))


def computeAAEAUC(output, target):
    """utils.computeAAEAUC (utils.py:96-140): AAE (deg, 60-degree field of view over 224 px) and the
    single-threshold AUC proxy.  numpy arrays (what the reference's callers pass) are evaluated on the host with
    scipy like the reference; CUDA tensors (what this package's drivers pass) go through the device kernel."""
    if isinstance(output, torch.Tensor) and output.is_cuda and tuple(output.shape[-2:]) == (224, 224):
        return _aae_auc_device(output, target)          # the metric's constants (224, 112) are the reference's
    if isinstance(output, torch.Tensor):
        output, target = output.detach().cpu().numpy().squeeze(), target.detach().cpu().numpy().squeeze()
    if output.ndim == 3:
        aae, auc, gp = [], [], []
        for b in range(output.shape[0]):
            a, u, p = _aae_auc_one(output[b].squeeze(), target[b].squeeze(), output.shape[2] * output.shape[1])
            aae.append(a); auc.append(u); gp.append(p)
        return np.mean(aae), np.mean(auc), gp
    a, u, p = _aae_auc_one(output, target, output.shape[0] * output.shape[1])
    return a, u, [p]


def plot_loss(train_loss, test_loss, save_path):
    try:
        import matplotlib
        matplotlib.use('agg')
        import matplotlib.pyplot as plt
    except Exception:       # plotting is cosmetic; never fail a training run on it
        return
    plt.plot(train_loss)
    plt.plot(test_loss)
    plt.ylabel('loss')
    plt.xlabel('epoch')
    plt.legend(['train', 'test'], loc='upper right')
    plt.savefig(save_path)
    plt.close()
