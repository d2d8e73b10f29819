"""Turn a rocprofv3 output directory (csv or rocpd sqlite) into the text summary committed under profiles/.
Usage: python tools/prof_summary.py <rocprof_out_dir> <profiles/out.txt> "<command line that was profiled>" """
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_csv(d):
    rows = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = rows[r["Kernel_Name"]]
            k[0] += 1
            k[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    return rows


def from_db(d):
    rows = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(f).cursor()
        for name, calls, total in cur.execute("select name, count(*), sum(end-start)/1000.0 from kernels group by name"):
            rows[name][0] += calls
            rows[name][1] += total
    return rows


def main():
    d, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = from_csv(d) or from_db(d)
    tot = sum(v[1] for v in rows.values())
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n# MI355X (gfx950); durations in microseconds; total kernel time {tot/1000:.2f} ms\n")
        f.write("%-104s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, (calls, total) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            f.write("%-104s %8d %14.1f %12.1f %7.2f\n" % (name[:104], calls, total, total / calls, 100 * total / tot))
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main()
