"""Where does the step go when no matrix-core kernel is running?  Reads a rocprofv3 --kernel-trace CSV directory and, over
the steady-state window (last `--steps` bench steps), reports the time during which at least one MFMA-bound kernel
(conv fwd / dgrad / wgrad) is active, and attributes the rest (exposed time) to the kernels running then.
Usage: python tools/trace_mfma.py <rocprof_out_dir> [--skip-frac 0.5]"""
import csv
import glob
import os
import sys
from collections import defaultdict

MFMA = ("igemm", "wgrad9", "wgrad_ups", "conv3x3_wgrad", "conv_first")


def main():
    d = sys.argv[1]
    skip = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[2] == "--skip-frac" else 0.5
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    marks = [e[0] for e in ev if "floss_centroid" in e[2]]          # one per SP step
    if len(marks) >= 7:
        lo, hi = marks[-6], marks[-1]                # five whole steady-state steps
        nsteps = 5
    else:
        t0, t1 = ev[0][0], max(e[1] for e in ev)
        lo, hi, nsteps = t0 + (t1 - t0) * skip, t1, 0
    ev = [(max(s, lo), min(e, hi), n) for s, e, n in ev if e > lo and s < hi]
    pts = []
    for s, e, n in ev:
        m = any(k in n for k in MFMA)
        pts.append((s, 1, m, n))
        pts.append((e, -1, m, n))
    pts.sort(key=lambda p: (p[0], -p[1]))
    active, nm = defaultdict(int), 0
    last = pts[0][0]
    tot = mfma_t = idle = 0
    exposed = defaultdict(float)
    for t, dlt, m, n in pts:
        dt = t - last
        if dt > 0:
            tot += dt
            if nm > 0:
                mfma_t += dt
            elif not active:
                idle += dt
            else:
                share = dt / len(active)
                for k in active:
                    exposed[k] += share
        last = t
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")
        short = short.split("(")[0][:60]
        if dlt > 0:
            active[short] += 1
            nm += m
        else:
            active[short] -= 1
            if active[short] == 0:
                del active[short]
            nm -= m
    if nsteps:
        print(f"{nsteps} steps, {tot/1e6/nsteps:.2f} ms per step")
    print(f"window {tot/1e6:.2f} ms: MFMA-bound kernel active {100*mfma_t/tot:.1f} %, GPU idle {100*idle/tot:.1f} %, "
          f"only other kernels {100*(tot-mfma_t-idle)/tot:.1f} %")
    for k, v in sorted(exposed.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  exposed {v/1e6:8.3f} ms ({100*v/tot:4.1f} %)  {k}")


if __name__ == "__main__":
    main()
