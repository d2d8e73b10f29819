"""One-off probe: how does the CPU oracle's SP train step scale with torch threads on the GPU box's host?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import egaze_oracle as O, synth
threads = int(sys.argv[1]); batch = int(sys.argv[2])
torch.set_num_threads(threads)
sd = synth.synth_state_dict(O.sp_shapes(), seed=1, head_gain=0.25)
x_s, x_t, gt, _ = synth.synth_sp_batch(batch, 224, seed=0)
opt = {}
t0 = time.perf_counter(); O.sp_train_step(sd, opt, 1, x_s, x_t, gt, 1e-7); t1 = time.perf_counter()
O.sp_train_step(sd, opt, 2, x_s, x_t, gt, 1e-7); t2 = time.perf_counter()
print(f"threads={threads} batch={batch} warm={t1-t0:.2f}s timed={t2-t1:.2f}s fps={batch/(t2-t1):.2f}", flush=True)
