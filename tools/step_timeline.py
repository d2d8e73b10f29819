"""Timeline of ONE timed SP step from a rocprofv3 --kernel-trace database: every kernel of the forward phase longer than a threshold with
its start offset, duration and HIP stream (are the two encoder streams fed at the same time?), then per-stream busy time and the share
of the backward phase with an MFMA-bound kernel active.
Usage: rocprofv3 --kernel-trace -d /tmp/t -o sp -- python bench.py --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-f32-leg
       --no-roofline --no-at ; python tools/step_timeline.py /tmp/t/sp_results.db [min_us]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select d.start, d.end, s.kernel_name, d.stream_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()


def nm(n):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n).replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"[(<].*", "", n)[:34]


def union(iv):
    iv = sorted(iv)
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


big = [i for i, r in enumerate(rows) if ('adam_kernel' in r[2] or 'adam_dev' in r[2]) and r[1] - r[0] > 100e3]      # the SP optimizer's Adam
a0, a1 = big[-3], big[-2]
t0 = rows[a0][1]
seg = rows[a0 + 1:a1 + 1]
fend = next(i for i, r in enumerate(seg) if 'floss' in r[2] or 'bce' in r[2])
print(f"step span {(rows[a1][1] - t0) / 1e6:.2f} ms: forward {(seg[fend][0] - t0) / 1e6:.2f} ms, backward + Adam {(rows[a1][1] - seg[fend][0]) / 1e6:.2f} ms")
print(f"forward phase, kernels >= {thr:.0f} us (start us | duration | stream | kernel):")
for s, e, n, st in seg[:fend]:
    if e - s >= thr * 1e3:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} s{st} {nm(n)}")
MF = ("igemm", "wgrad9", "wgrad_ups", "conv_first")
for name, part in (("forward", seg[:fend]), ("backward", seg[fend:])):
    span = (part[-1][1] - part[0][0]) / 1e3
    mf = [(s, e) for s, e, n, st in part if any(k in n for k in MF)]
    print(f"{name}: span {span:.0f} us, some kernel active {union([(s, e) for s, e, *_ in part]) / 1e3:.0f} us, an MFMA-bound kernel active "
          f"{union(mf) / 1e3:.0f} us ({100 * union(mf) / 1e3 / span:.1f} %); per stream busy: "
          + ", ".join(f"s{st} {sum(e - s for s, e, n, x in part if x == st) / 1e3:.0f}" for st in sorted(set(r[3] for r in part))))
