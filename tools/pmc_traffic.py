"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic for one kernel family.
Usage: python tools/pmc_traffic.py <dir_fetch> <dir_write> <kernel substring[,substring...]> <out.json>
Counter units on gfx950 (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) coalesced read stream -> it is doubled here; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import sys


def total(d, counter, sub):
    tot, disp = 0.0, set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(t in r["Kernel_Name"] for t in sub.split(",")) and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
                disp.add(r["Dispatch_Id"])
    return tot, len(disp)


def main():
    dfetch, dwrite, sub, out = sys.argv[1:5]
    f, nf = total(dfetch, "FETCH_SIZE", sub)
    w, nw = total(dwrite, "WRITE_SIZE", sub)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res = {"kernel": sub, "launches": nf, "kernel_src_sha": bench.kernel_src_sha(), "fetch_kib_raw_per_launch": f / max(nf, 1),
           "write_kib_per_launch": w / max(nw, 1),
           "traffic_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
           "note": "FETCH_SIZE doubled (gfx950 half-count of wide coalesced reads), WRITE_SIZE uncorrected; "
                   "rocprofv3 --pmc in separate passes"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
