"""Summarise a rocprofv3 --hip-trace --kernel-trace output directory of tools/dp_world1.py --trace: HIP API calls per name
and kernels per name (what a bucket hand-over costs in events / waits / launches).  Usage: python tools/dp_trace_summary.py <dir>"""
import csv
import glob
import os
import sys
from collections import Counter, defaultdict

d = sys.argv[1]
api, kern = Counter(), defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        api[r["Function"]] += 1
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = kern[r["Kernel_Name"]]
        k[0] += 1
        k[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
print("HIP API calls:")
for n, c in api.most_common(25):
    print(f"  {c:8d}  {n}")
print("kernels matching nccl / rccl / copy / fill:")
for n, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][1]):
    if any(s in n.lower() for s in ("nccl", "rccl", "copybuffer", "fillbuffer", "allreduce", "ncclDev")):
        print(f"  {c:6d} calls {t:10.1f} us total {t / c:8.1f} us avg  {n[:100]}")
