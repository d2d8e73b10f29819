"""ISA statistics of the streamed conv kernel variants from a device assembly dump (hipcc -S --cuda-device-only):
   python tools/isa_stats.py file.s [filter e.g. 'f16,1,0']   -> per kernel: instruction counts of the main opcodes."""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
pat = re.compile(r'^(_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)I(DF16_|DF16b)((?:L[ib]\d+E)+)E[^\n:]*):', re.M)
ms = list(pat.finditer(s))
keys = ['v_mfma_f32_32x32x16_f16', 'v_mfma_f32_32x32x16_bf16', 'ds_read_b128', 'ds_write_b64', 'buffer_load_dwordx4', 's_barrier', 's_waitcnt',
        's_nop', 'scratch_load_dword', 'scratch_store_dword', 'scratch_load_dwordx4', 'scratch_store_dwordx4', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32']
for i, m in enumerate(ms):
    name = m.group(2) + "<" + ("f16" if m.group(3) == "DF16_" else "bf16") + "," + ",".join(re.findall(r'L[ib](\d+)E', m.group(4))) + ">"
    if flt not in name:
        continue
    body = s[m.end(): s.index('.Lfunc_end', m.end())]
    cnt = {}
    n = 0
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        n += 1
        op = l.split()[0]
        cnt[op] = cnt.get(op, 0) + 1
    print(name, "instrs", n, {k.replace('v_mfma_f32_32x32x16_', 'mfma_'): cnt[k] for k in keys if cnt.get(k)})
