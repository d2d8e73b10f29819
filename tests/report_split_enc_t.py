"""Debug report: ReLU decisions that differ between the split-half and exact-f32 modes in the temporal encoder output."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_hip_model_sp as T
import egaze_amd.hipops as H
size = 32
x_s, x_t, gt, _ = T.synth.synth_sp_batch(3, size, seed=5)
outs = {}
for mode in ("f32", "split"):
    H.PRECISION = mode
    model, sd0 = T.build_model(); model.train()
    for which, enc, inp in (("t", model.features_t, x_t), ("s", model.features_s, x_s)):
        outs[(mode, which)] = enc(inp.to(T.DEV)).detach().float().cpu()
for which in ("t", "s"):
    a, b = outs[("f32", which)], outs[("split", which)]
    flip = (a > 0) != (b > 0)
    print(which, "elements", a.numel(), "ReLU decisions that differ:", int(flip.sum()), "values there:", a[flip].tolist(), b[flip].tolist(),
          "max abs diff %.2e" % (a - b).abs().max().item())
