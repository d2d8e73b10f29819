"""Gradient accuracy of one SP train step at 224 x 224 (B = 2, the s224 golden inputs) against an fp64 run of the CPU
oracle: norm deviation and element-wise L2 error per parameter for the HIP path (EGAZE_PRECISION=split|f32), the golden
fixture (reference, torch CPU fp32) and the fp32 oracle.  Test infrastructure (imports oracle/): run on a GPU box."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_hip_model_sp as T
from egaze_amd.floss import floss
from egaze_amd.optim import FusedAdam
gold = np.load(os.path.join(T.GOLDEN, "model_sp_s224.npz"))
for rep in range(2):
    model, sd0 = T.build_model()
    x_s, x_t, gt, _ = T.synth.synth_sp_batch(2, 224, seed=0)
    model.train()
    out = model(x_s.to(T.DEV), x_t.to(T.DEV))
    loss = floss().to(T.DEV)(out, gt.to(T.DEV).view(out.size()))
    loss.backward()
    keys = [k[5:] for k in gold.files if k.startswith("gsum/")]
    worst = []
    floor = 1e-5 * max(gold["gsum/" + k][0] for k in keys)
    for k, p in model.named_parameters():
        want = gold["gsum/" + k][0]; got = p.grad.double().norm().item()
        worst.append((abs(got - want) / (want + 50 * floor), k, got, want))
    worst.sort(reverse=True)
    print(os.environ.get("EGAZE_PRECISION"), os.environ.get("EGAZE_STREAMS"), "loss", loss.item(), [(f"{w[0]:.2e}", w[1]) for w in worst[:6]])
    if rep == 0:
        import time
        from oracle import egaze_oracle as O
        torch.set_num_threads(32)
        t0 = time.time()
        w64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        _, out64, g64 = O.sp_train_step(w64, {}, 1, x_s.double(), x_t.double(), gt.double(), 0.0)
        print("fp64 oracle took", time.time() - t0)
        w32 = {k: v.clone() for k, v in sd0.items()}
        _, out32, g32 = O.sp_train_step(w32, {}, 1, x_s, x_t, gt, 0.0)
        rows = []
        for k, p in model.named_parameters():
            n64 = g64[k].norm().item()
            if n64 < 50 * floor: continue
            nh = p.grad.double().norm().item(); ng = gold["gsum/" + k][0]
            eh = (p.grad.double().cpu() - g64[k]).norm().item() / n64
            ec = (g32[k].double() - g64[k]).norm().item() / n64
            rows.append((abs(nh - n64) / n64, abs(ng - n64) / n64, eh, ec, k))
        rows.sort(reverse=True)
        print("norm dev HIP-vs-fp64 | golden(CPU fp32)-vs-fp64 | elementwise L2 err HIP-vs-fp64")
        for r in rows[:12]: print("%.2e %.2e %.2e %.2e %s" % r)
        import numpy as np
        a = np.array([[r[0], r[1], r[2], r[3]] for r in rows]); print("median", np.median(a, 0), "max", a.max(0))
        break
