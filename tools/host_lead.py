"""How far ahead of the device is the host?  From a rocprofv3 --hip-trace --kernel-trace database: for every kernel of one timed SP
step, the time between the END of its launch call on the host and the START of its execution on the device, per stream.
Usage (on the GPU box): rocprofv3 --hip-trace --kernel-trace -d /tmp/t -o sp -- python bench.py --steps 3 --warmup 2 --repeats 1 ...
                        python tools/host_lead.py /tmp/t/sp_results.db"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
def cols(t):
    return [r[1] for r in c.execute(f"pragma table_info({t})")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
reg = [t for t in tabs if t.startswith('rocpd_region')]
if "--schema" in sys.argv:
    for t in tabs:
        print(t, cols(t))
    sys.exit(0)
reg = reg[0]
strs = [t for t in tabs if t.startswith('rocpd_string')][0]
ev = [t for t in tabs if t.startswith('rocpd_event')][0]
# kernel dispatches with their event's correlation id
q = f"""select d.start, d.end, s.kernel_name, d.stream_id, e.correlation_id
        from {kd} d join {ks} s on d.kernel_id = s.id join {ev} e on d.event_id = e.id order by d.start"""
rows = c.execute(q).fetchall()
# API regions (hipLaunchKernel etc.) by correlation id
q2 = f"""select e.correlation_id, r.start, r.end, st.string from {reg} r join {ev} e on r.event_id = e.id join {strs} st on r.name_id = st.id
         where st.string like 'hip%Launch%' or st.string like 'hipModuleLaunch%' or st.string like 'hipExtLaunch%'"""
api = {}
for cid, s, e, n in c.execute(q2):
    api[cid] = (s, e, n)
big = [i for i, r in enumerate(rows) if ('adam_kernel' in r[2] or 'adam_dev' in r[2]) and r[1] - r[0] > 100e3]
a0, a1 = big[-3], big[-2]
t0 = rows[a0][1]
seg = rows[a0 + 1:a1 + 1]
def nm(n):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n).replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"[(<].*", "", n)[:36]
print(f"step span {(rows[a1][1] - t0) / 1e6:.2f} ms, {len(seg)} kernels; columns: device start (us from the previous Adam's end) | duration | stream | "
      "host lead = device start - end of the launch call (us; negative = the device waited for the host) | kernel")
leads = {}
for i, (s, e, n, st, cid) in enumerate(seg):
    a = api.get(cid)
    lead = (s - a[1]) / 1e3 if a else float('nan')
    leads.setdefault(st, []).append(lead)
    if i < 60 or i % 12 == 0:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} s{st} lead {lead:9.1f}  {nm(n)}")
for st, v in leads.items():
    v = [x for x in v if x == x]
    v.sort()
    print(f"stream {st}: {len(v)} kernels, host lead min {v[0]:.0f} / median {v[len(v) // 2]:.0f} / max {v[-1]:.0f} us")
