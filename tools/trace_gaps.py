"""GPU busy fraction and idle gaps from a rocprofv3 --kernel-trace csv: union of kernel intervals over the traced window
(last `frac` of the run = steady-state steps).  Usage: python tools/trace_gaps.py <rocprof_out_dir> [frac]"""
import csv, glob, os, sys
d = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
iv.sort()
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
w0 = t1 - (t1 - t0) * frac
iv = [x for x in iv if x[0] >= w0]
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - iv[0][0]
print(f"window {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms = {100*busy/span:.1f} %, {len(iv)} kernels, {len(gaps)} gaps, "
      f"gap total {sum(g for g,_ in gaps)/1e6:.2f} ms, gaps > 20us: {sum(1 for g,_ in gaps if g > 20000)} "
      f"totalling {sum(g for g,_ in gaps if g > 20000)/1e6:.2f} ms")
gaps.sort(reverse=True)
for g, n in gaps[:12]:
    print(f"  {g/1e3:8.1f} us before {n[:90]}")
