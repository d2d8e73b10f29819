"""Element-wise gradient agreement of one SP train step at the HEADLINE geometry (B = 32, 224 x 224, seed 3 -- the inputs of
test_model_sp_train_step_headline_size): per parameter tensor, the share of entries within 2e-3 of max |ref| (mostly_close)
for HIP vs the fp32 CPU oracle, and -- with an fp64 run of the oracle as the truth -- HIP vs fp64 and CPU fp32 vs fp64.
Test infrastructure (imports oracle/): run on a GPU box.  Usage: python tests/report_headline_grads.py [--no-f64] [--batch B]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_hip_model_sp as T
from egaze_amd.floss import floss
from oracle import egaze_oracle as O

f64 = "--no-f64" not in sys.argv
BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 32
torch.set_num_threads(min(32, os.cpu_count() or 1))
model, sd0 = T.build_model()
x_s, x_t, gt, _ = T.synth.synth_sp_batch(BATCH, 224, seed=3)
print('batch', BATCH)
model.train()
out = model(x_s.to(T.DEV), x_t.to(T.DEV))
floss().to(T.DEV)(out, gt.to(T.DEV).view(out.size())).backward()
hip = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
del out
t0 = time.time()
_, _, g32 = O.sp_train_step({k: v.clone() for k, v in sd0.items()}, {}, 1, x_s, x_t, gt, 0.0)
print("fp32 oracle step %.0f s" % (time.time() - t0), flush=True)
g64 = None
if f64:
    t0 = time.time()
    w64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    _, _, g64 = O.sp_train_step(w64, {}, 1, x_s.double(), x_t.double(), gt.double(), 0.0)
    print("fp64 oracle step %.0f s" % (time.time() - t0), flush=True)
gabs = max(g.abs().max().item() for g in g32.values())
print("%-28s %9s %9s | %9s %9s | %9s %9s" % ("tensor", "cos h/32", "frac h/32", "frac h/64", "frac 32/64", "L2 h/64", "L2 32/64"))
rows = []
for k, ref in g32.items():
    if ref.abs().max().item() < 1e-5 * gabs:
        continue
    c, _ = T.cos_norm(hip[k].numpy(), ref.numpy())
    _, f_h32 = T.mostly_close(hip[k].numpy(), ref.numpy())
    f_h64 = f_c64 = l_h = l_c = float("nan")
    if g64 is not None:
        t = g64[k].numpy()
        _, f_h64 = T.mostly_close(hip[k].numpy(), t)
        _, f_c64 = T.mostly_close(ref.numpy(), t)
        n = np.linalg.norm(t)
        l_h = np.linalg.norm(hip[k].double().numpy() - t) / n
        l_c = np.linalg.norm(ref.double().numpy() - t) / n
    rows.append((k, c, f_h32, f_h64, f_c64, l_h, l_c))
    print("%-28s %9.6f %9.4f | %9.4f %9.4f | %9.2e %9.2e" % rows[-1], flush=True)
a = np.array([r[1:] for r in rows])
print("min over tensors:", np.nanmin(a, 0))
print("tensors below 0.98 (HIP vs fp32 oracle):", [(r[0], round(r[2], 4), round(r[3], 4), round(r[4], 4)) for r in rows if r[2] < 0.98])
