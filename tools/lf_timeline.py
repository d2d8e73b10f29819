"""Timeline of ONE LF training step from a rocprofv3 kernel trace: start offset, duration, queue, kernel (critical-path reading).
Usage: python tools/lf_timeline.py <rocprof_out_dir> [step_index_from_end]"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
ev.sort()
marks = [i for i, e in enumerate(ev) if "floss_centroid" in e[2]]
lo = marks[back]                      # `back`-th step from the START of the trace (the eager legs come first)
# a step = from the first kernel after the previous step's adam to this step's adam
starts = [i for i, e in enumerate(ev) if "adam_kernel" in e[2] or "adam_dev_kernel" in e[2]]
a0 = max(i for i in starts if i < lo)
a1 = min(i for i in starts if i > lo)
t0 = ev[a0][1]
busy = 0
for s, e, n, q in ev[a0 + 1:a1 + 1]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  q{q:>3}  {n[:90]}")
print(f"step span {(ev[a1][1] - t0) / 1e3:.1f} us, {a1 - a0} kernels")
