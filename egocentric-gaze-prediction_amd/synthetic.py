"""Synthetic batches with the tensor contracts of the reference's datasets (no disk, no JPEG decode):
STDataset  (data/STdatas.py:50-73):  image (B,3,H,W) = (u8/255 - mean)/std on BGR-ordered channels,
                                     flow (B,20,H,W) = (u8/255 - 0.5)/0.5 (x_t,y_t,...,x_{t-9},y_{t-9}),
                                     gt (B,1,H,W) = uint8-quantised Gaussian blob / 255
lstmDataset (data/LSTMdatas.py:59):  input / gt 512-vectors (spatial means of post-ReLU features, >= 0)
lateDataset (data/lateDataset.py:22-33): im / feat / gt = u8/255 maps (B,1,H,W)
Generated directly on the device with a seeded torch generator (bench / smoke inputs)."""
import torch

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def gaze_maps(batch, size, device, g):
    lo, hi = (20.0, size - 20.0) if size > 60 else (2.0, size - 2.0)
    c = torch.rand(batch, 2, device=device, generator=g) * (hi - lo) + lo
    r = torch.arange(size, device=device, dtype=torch.float32)
    sr, sc = 16.3 * size / 224.0, 12.25 * size / 224.0
    gr = torch.exp(-((r[None, :] - c[:, 0:1]) ** 2) / (2 * sr * sr))
    gc = torch.exp(-((r[None, :] - c[:, 1:2]) ** 2) / (2 * sc * sc))
    m = gr[:, :, None] * gc[:, None, :]
    return (torch.round(m * 255.0) / 255.0).unsqueeze(1).contiguous()


def sp_batch(batch, size=224, device="cuda", seed=0):
    g = _gen(device, seed)
    img = torch.randint(0, 256, (batch, 3, size, size), device=device, generator=g).float() / 255.0
    mean = torch.tensor(_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(_STD, device=device).view(1, 3, 1, 1)
    image = ((img - mean) / std).contiguous()
    flow = ((torch.randint(0, 256, (batch, 20, size, size), device=device, generator=g).float() / 255.0 - 0.5)
            / 0.5).contiguous()
    gt = gaze_maps(batch, size, device, g)
    fixsac = (torch.rand(batch, 1, device=device, generator=g) < 0.746).float()
    return {"image": image, "flow": flow, "gt": gt, "fixsac": fixsac}


def at_batch(T, B, device="cuda", seed=0):
    g = _gen(device, seed)
    inp = torch.randn(T, B, 512, device=device, generator=g).abs() * 0.5
    gt = torch.randn(T, B, 512, device=device, generator=g).abs() * 0.5
    return {"input": inp, "gt": gt}


def lf_batch(batch, size=224, device="cuda", seed=0):
    g = _gen(device, seed)
    im = torch.randint(0, 256, (batch, 1, size, size), device=device, generator=g).float() / 255.0
    feat = torch.randint(0, 256, (batch, 1, size, size), device=device, generator=g).float() / 255.0
    return {"im": im, "feat": feat, "gt": gaze_maps(batch, size, device, g)}
