"""TEST INFRASTRUCTURE ONLY -- deterministic synthetic weights and inputs.

Weights are never committed (the SP model is 186 MB): both the golden generator
(reference side, ``tests/golden/make_golden.py``) and the tests (oracle / HIP side)
regenerate them from ``np.random.RandomState`` seeded per state-dict key, so the
result does not depend on key iteration order.

Input contracts follow data/STdatas.py:50-73 and data/lateDataset.py:22-33
(SURVEY.md section 8d).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np
import torch


def _rs(key: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 32))


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 1,
                     head_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Fill every entry of a state-dict-shaped mapping deterministically.

    conv / linear / lstm weights: N(0, sqrt(2/fan_in)) (keeps activations O(1));
    biases: N(0, 0.05); BN gamma U(0.5,1.5), beta N(0,0.1), running_mean N(0,0.1),
    running_var U(0.5,1.5), num_batches_tracked 0.
    """
    bn_prefixes = {k[: -len("running_mean")] for k in shapes if k.endswith("running_mean")}
    out = {}
    for k, shp in shapes.items():
        rs = _rs(k, seed)
        shp = tuple(shp)
        prefix = k[: k.rfind(".") + 1]
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.int64)
            continue
        if k.endswith("running_mean"):
            a = rs.standard_normal(shp) * 0.1
        elif k.endswith("running_var"):
            a = rs.uniform(0.5, 1.5, shp)
        elif prefix in bn_prefixes and k.endswith("weight"):
            a = rs.uniform(0.5, 1.5, shp)
        elif prefix in bn_prefixes and k.endswith("bias"):
            a = rs.standard_normal(shp) * 0.1
        elif len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            a = rs.standard_normal(shp) * np.sqrt(2.0 / fan_in)
            if shp[0] == 1:                       # 1x1 heads: keep the sigmoid un-saturated
                a = a * head_gain
        else:
            a = rs.standard_normal(shp) * 0.05
        out[k] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


_MEAN = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)


def synth_sp_batch(batch: int, size: int = 224, seed: int = 0):
    """(image, flow, gt, fixsac) as STDataset yields them (data/STdatas.py:50-73).

    image: (u8/255 - mean)/std on BGR-ordered channels; flow: (u8/255 - 0.5)/0.5 for the
    10-pair stack (20 ch); gt: uint8-quantised anisotropic Gaussian (sigma 16.3 x 12.25 px,
    data/dataset_preprocessing.py:113-119) / 255; fixsac in {0,1} with P(1)=0.746."""
    rs = np.random.RandomState(seed)
    img_u8 = rs.randint(0, 256, (batch, 3, size, size)).astype(np.float32)
    image = (img_u8 / 255.0 - _MEAN) / _STD
    flow_u8 = rs.randint(0, 256, (batch, 20, size, size)).astype(np.float32)
    flow = (flow_u8 / 255.0 - 0.5) / 0.5
    gt = synth_gt(batch, size, rs)
    fixsac = (rs.uniform(size=(batch, 1)) < 0.746).astype(np.float32)
    return (torch.from_numpy(image.astype(np.float32)), torch.from_numpy(flow.astype(np.float32)),
            torch.from_numpy(gt), torch.from_numpy(fixsac))


def synth_gt(batch: int, size: int, rs: np.random.RandomState) -> np.ndarray:
    lo, hi = (20, size - 20) if size > 60 else (2, size - 2)
    cr = rs.uniform(lo, hi, batch)
    cc = rs.uniform(lo, hi, batch)
    sr, sc = 16.3 * size / 224.0, 12.25 * size / 224.0
    r = np.arange(size, dtype=np.float64)[None, :, None]
    c = np.arange(size, dtype=np.float64)[None, None, :]
    g = np.exp(-((r - cr[:, None, None]) ** 2 / (2 * sr ** 2) + (c - cc[:, None, None]) ** 2 / (2 * sc ** 2)))
    g = np.round(255.0 * g) / 255.0
    return g.astype(np.float32)[:, None]


def synth_lf_batch(batch: int, size: int = 224, seed: int = 0):
    """(im, feat, gt) as lateDataset yields them: u8/255 maps (data/lateDataset.py:22-33)."""
    rs = np.random.RandomState(seed)
    im = rs.randint(0, 256, (batch, 1, size, size)).astype(np.float32) / 255.0
    feat = rs.randint(0, 256, (batch, 1, size, size)).astype(np.float32) / 255.0
    gt = synth_gt(batch, size, rs)
    return torch.from_numpy(im), torch.from_numpy(feat), torch.from_numpy(gt)


def synth_at_batch(T: int, B: int, seed: int = 0):
    """AT inputs/targets: spatial means of post-ReLU features (>= 0), shape (T,B,512)."""
    rs = np.random.RandomState(seed)
    inp = np.abs(rs.standard_normal((T, B, 512))).astype(np.float32) * 0.5
    tgt = np.abs(rs.standard_normal((T, B, 512))).astype(np.float32) * 0.5
    return torch.from_numpy(inp), torch.from_numpy(tgt)
