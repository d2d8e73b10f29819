"""BASELINE config 5 timing on one GPU: the stages of the full SP -> AT -> LF pipeline on synthetic frames held in memory.
  1. SP training step (the headline, B = 32)                           -> see bench.py
  2. SP inference as AT.extract_late runs it (batch 1) and at batch 32
  3. AT.extract_late per frame (SP forward + device metric + crop mean + LSTM step + weighted map + uint8 hand-over),
     with the PNG writes replaced by a no-op (disk I/O is the reference's, not the path's)
  4. LF training step (B = 32)
Usage: python tools/bench_pipeline.py [--frames 64]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egaze_amd  # noqa
import egaze_amd.AT as at_mod
from egaze_amd.AT import AT
from egaze_amd.floss import floss
from egaze_amd.models.late_fusion import late_fusion
from egaze_amd.models.model_SP import model_SP
from egaze_amd.optim import FusedAdam
from egaze_amd.utils import cfg, make_layers
from egaze_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda", 0)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


torch.manual_seed(0)
sp = model_SP(make_layers(cfg['D'], 3), make_layers(cfg['D'], 20)).to(dev).eval()
for B in (1, 32):
    b = synthetic.sp_batch(B, 224, dev, seed=1)
    with torch.no_grad():
        dt = timed(lambda: sp(b["image"], b["flow"]), 20 if B == 1 else 5)
    print(f"SP inference B={B}: {dt*1e3:.2f} ms per batch, {B/dt:.0f} frames/s")

with tempfile.TemporaryDirectory() as d:
    torch.save({'state_dict': sp.state_dict()}, os.path.join(d, "sp.pth.tar"))
    for sub in ("train", "test"):
        os.makedirs(os.path.join(d, "512w", sub))
        for i in range(2):
            torch.save(torch.zeros(512), os.path.join(d, "512w", sub, f"fix_v_{i:010d}.pth.tar"))
    at = AT(pretrained_model=os.path.join(d, "sp.pth.tar"), save_path=d, device='0', lstm_data_path=os.path.join(d, "512w"))
    b = synthetic.sp_batch(a.frames, 224, torch.device("cpu"), seed=2)
    rs = np.random.RandomState(0)
    loader = [{"imname": ["f%05d.png" % i], "fixsac": torch.tensor([[float(rs.rand() < 0.746)]]),
               "image": b["image"][i:i + 1], "flow": b["flow"][i:i + 1], "gt": b["gt"][i:i + 1]} for i in range(a.frames)]
    at_mod.imwrite = lambda path, arr: None                    # disk writes are not part of the path
    at_mod._progress = lambda it: it
    at_mod.resize = lambda arr, size: arr                       # (the x16 resize of the 14 x 14 map is host image I/O like the PNG write)
    rs8 = np.random.RandomState(1)
    loader_u8 = [{"imname": s["imname"], "fixsac": s["fixsac"],
                  "image": torch.from_numpy(rs8.randint(0, 256, (1, 3, 224, 224)).astype(np.uint8)),
                  "flow": torch.from_numpy(rs8.randint(0, 256, (1, 20, 224, 224)).astype(np.uint8)),
                  "gt": (s["gt"] * 255).to(torch.uint8)} for s in loader]
    # gaze_full.py hands extract_late a DataLoader(pin_memory=True): the in-memory stand-in pins its samples the same way
    for ld_ in (loader, loader_u8):
        for s in ld_:
            for k in ("image", "flow", "gt"):
                s[k] = s[k].contiguous().pin_memory()
    for tag, ld in (("fp32 frames (the reference loader's format)", loader), ("uint8 frames (this build's STDataset(raw_u8=True))", loader_u8)):
        at.extract_late(ld[:33], d + "/p/", d + "/f/")
        dts = []
        for _ in range(3):                                        # the steady state of a long extraction: best of three passes
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            at.extract_late(ld, d + "/p/", d + "/f/")
            torch.cuda.synchronize()
            dts.append((time.perf_counter() - t0) / a.frames)
        print("   passes, ms per frame:", [round(v * 1e3, 3) for v in dts])
        dt = min(dts)
        at.extract_profile = {}
        at.extract_late(ld, d + "/p/", d + "/f/")
        print("   phases, ms per frame (each phase synchronised):", {k: round(v / a.frames * 1e3, 3) for k, v in at.extract_profile.items()})
        at.extract_profile = None
        print(f"AT.extract_late, {tag} in host memory, no PNG writes / resize: {dt*1e3:.2f} ms per frame, {1/dt:.0f} frames/s")

lf = late_fusion().to(dev).train()
crit = floss().to(dev)
opt = FusedAdam(lf.parameters(), lr=1e-4)
im, feat, gt = (torch.rand(32, 1, 224, 224, device=dev) for _ in range(3))


def lf_step():
    out = lf(feat, im)
    loss = crit(out, gt)
    opt.zero_grad()
    loss.backward()
    opt.step()


dt = timed(lf_step, 20)
print(f"LF train step B=32: {dt*1e3:.2f} ms, {32/dt:.0f} frames/s")
