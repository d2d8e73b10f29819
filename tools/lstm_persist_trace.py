"""Where a global step of the persistent LSTM kernels goes: a variant build with -DEGZ_PERSIST_TRACE (thread 0 of block (0, 0) stamps
the 100 MHz wall clock at the phase boundaries of every step), read back from the tail of the sync scratch.
    EGZ_VARIANT=trace bash egocentric-gaze-prediction_amd/csrc/build.sh -DEGZ_PERSIST_TRACE
    EGAZE_HIP_LIB=egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_trace.so python tools/lstm_persist_trace.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import egaze_amd  # noqa: E402,F401
from egaze_amd import hipops as H  # noqa: E402

T, B = 16, 32
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
w_ih = [None, (torch.randn(2048, 512, generator=g) * 0.05).to(dev)]
w_hh = [(torch.randn(2048, 512, generator=g) * 0.05).to(dev) for _ in range(2)]
b_ih = [(torch.randn(2048, generator=g) * 0.1).to(dev) for _ in range(2)]
b_hh = [(torch.randn(2048, generator=g) * 0.1).to(dev) for _ in range(2)]
bsum = [a_ + b_ for a_, b_ in zip(b_ih, b_hh)]
w_hh_t = [w.t().contiguous() for w in w_hh]
w_ih_t = [None, w_ih[1].t().contiguous()]
gx0 = torch.randn(T, B, 2048, generator=g).to(dev)
h0 = (torch.randn(2, B, 512, generator=g) * 0.5).to(dev)
c0 = (torch.randn(2, B, 512, generator=g) * 0.5).to(dev)
dh_top = torch.randn(T, B, 512, generator=g).to(dev)
ref_f = H.lstm_wave_fwd(gx0 + bsum[0], w_ih, w_hh, bsum, h0, c0)
ref_b = H.lstm_wave_bwd(dh_top, None, None, ref_f[2], ref_f[1], c0, w_hh_t, w_ih_t)
NAMES = ["prefetch + wait (poll + barrier)", "sc1 loads + MFMA + LDS partials + barrier", "reduce + cell",
         "gather + write-through store + drain + barrier", "arrival atomic (issue)"]
for direction, nsteps in (("fwd", T + 1), ("bwd", T + 2)):
    acc = []
    for rep in range(12):
        if direction == "fwd":
            hs, cs, acts, hn, cn = H.lstm_persist_fwd(gx0, w_ih, w_hh, b_ih, b_hh, h0, c0)
        else:
            got_b = H.lstm_persist_bwd(dh_top, None, None, acts, cs, c0, w_hh, w_ih, None)
        torch.cuda.synchronize()
        if direction == "fwd":
            worst = max(float((x - y).abs().max()) for x, y in zip((hs, cs, acts, hn, cn), ref_f))
        else:
            worst = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(got_b, ref_b))
        assert worst < 1e-4 and H.lstm_persist_status() == 0, (direction, rep, worst, H.lstm_persist_status())
        sync = next(iter(H._PERSIST_SYNC.values())).clone()
        st = sync[1280:1280 + 64 * 16].cpu().numpy().view(np.uint64).reshape(64, 8)[:nsteps, :6].astype(np.int64)
        if rep >= 2:
            acc.append(st)
    st = np.stack(acc)                                   # [rep][step][phase]
    d = np.diff(st, axis=2) * 10.0                        # ns
    nxt = (st[:, 1:, 0] - st[:, :-1, 5]) * 10.0
    total = (st[:, -1, 5] - st[:, 0, 0]) * 10.0
    print(f"--- {direction}: block (0,0), {st.shape[0]} launches, {nsteps} global steps, launch span {total.mean() / 1e3:.1f} us "
          f"= {total.mean() / nsteps / 1e3:.2f} us per step")
    mid = slice(2, nsteps - 2)
    for i, n in enumerate(NAMES):
        print(f"  {n:48s} {d[:, mid, i].mean() / 1e3:6.2f} us")
    print(f"  {'loop back':48s} {nxt[:, 2:-2].mean() / 1e3:6.2f} us")
