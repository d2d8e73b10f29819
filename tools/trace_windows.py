"""Chronological list of the windows of ONE steady-state SP step in which no MFMA-bound kernel is running (exposed HBM passes and
idle gaps), with what runs in them and which MFMA kernels border them.  Reads a rocprofv3 --kernel-trace CSV directory.
Usage: python tools/trace_windows.py <rocprof_out_dir> [min_us]"""
import csv, glob, os, sys

MFMA = ("igemm", "wgrad9", "wgrad_ups", "conv3x3_wgrad", "conv_first")
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")
    return n.split("(")[0].split("<")[0][:40]


marks = [e[0] for e in ev if "floss_centroid" in e[2]]
lo, hi = marks[-3], marks[-2]                      # one whole steady-state step (loss kernel to loss kernel)
ev = [(s, e, n) for s, e, n in ev if e > lo and s < hi]
mf = sorted((max(s, lo), min(e, hi)) for s, e, n in ev if any(k in n for k in MFMA))
# union of MFMA intervals
un = []
for s, e in mf:
    if un and s <= un[-1][1]:
        un[-1][1] = max(un[-1][1], e)
    else:
        un.append([s, e])
wins = []
cur = lo
for s, e in un:
    if s > cur:
        wins.append((cur, s))
    cur = max(cur, e)
if cur < hi:
    wins.append((cur, hi))
tot = sum(b - a for a, b in wins)
print(f"step {(hi - lo) / 1e6:.3f} ms (under the tracer); {len(wins)} windows without an MFMA-bound kernel, {tot / 1e6:.3f} ms in total")
print(f"windows >= {thr:.0f} us, in time order (offset from the loss kernel, duration, busy share, kernels inside):")
big = 0.0
for a, b in wins:
    if (b - a) / 1e3 < thr:
        continue
    big += b - a
    inside = [(max(s, a), min(e, b), n) for s, e, n in ev if e > a and s < b and not any(k in n for k in MFMA)]
    inside.sort()
    busy, ce = 0, a
    for s, e, n in inside:
        if e > ce:
            busy += e - max(s, ce)
            ce = e
    names = []
    for s, e, n in inside:
        nm = f"{short(n)}:{(e - s) / 1e3:.0f}"
        names.append(nm)
    prev = [short(n) for s, e, n in ev if any(k in n for k in MFMA) and abs(e - a) < 2000]
    print(f"  +{(a - lo) / 1e6:7.3f} ms  {(b - a) / 1e3:7.1f} us  busy {100 * busy / (b - a):3.0f} %  after {prev[:1]}  [{', '.join(names[:14])}{' ...' if len(names) > 14 else ''}]")
print(f"windows >= {thr:.0f} us: {big / 1e6:.3f} ms; smaller ones: {(tot - big) / 1e6:.3f} ms")
# the longest window kernel by kernel (start offset inside the window, duration, gap to the previous kernel's end)
a, b = max(wins, key=lambda w: w[1] - w[0])
print(f"longest window (+{(a - lo) / 1e6:.3f} ms, {(b - a) / 1e3:.1f} us), kernel by kernel:")
ce = a
for s_, e_, n in sorted((max(s, a), min(e, b), n) for s, e, n in ev if e > a and s < b):
    print(f"    +{(s_ - a) / 1e3:7.1f} us  {(e_ - s_) / 1e3:6.1f} us  gap {max(0, s_ - ce) / 1e3:6.1f}  {short(n)}")
    ce = max(ce, e_)
print(f"    +{(b - a) / 1e3:7.1f} us  (next MFMA-bound kernel starts; gap {max(0, b - ce) / 1e3:.1f})")
