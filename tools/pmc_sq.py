"""Average rocprofv3 --pmc counters per launch for kernels matching a substring.
Usage: python tools/pmc_sq.py <rocprof_out_dir> <kernel substring> [<kernel substring> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, subs = sys.argv[1], sys.argv[2:]
    for sub in subs:
        tot, disp = defaultdict(float), set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if sub in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
                    disp.add((f, r["Dispatch_Id"]))
        n = max(len(disp), 1)
        print(f"== {sub}: {len(disp)} launches")
        wc = tot.get("SQ_WAVE_CYCLES", 0.0)
        for k in sorted(tot):
            extra = f"  ({100 * tot[k] / wc:5.1f}% of SQ_WAVE_CYCLES)" if wc and k.startswith("SQ_") and k != "SQ_WAVE_CYCLES" else ""
            print(f"  {k:32s} {tot[k] / n:16.1f}{extra}")


if __name__ == "__main__":
    main()
