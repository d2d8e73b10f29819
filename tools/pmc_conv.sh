#!/bin/bash
# SQ / GRBM counters of the split-half conv kernels on one SP layer shape (tools/bench_conv.py), separate --pmc passes.
# Usage (GPU box): bash tools/pmc_conv.sh <out_prefix> <bench_conv args...>
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS"
P3="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  timeout 280 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- python $R/tools/bench_conv.py "$@" > /tmp/pmc$i.log 2>&1 || tail -5 /tmp/pmc$i.log
done
python $R/tools/pmc_sq.py /tmp/pmc1 igemm_x3 wgrad9_x3 wgrad_ups_x3 > $R/$OUT.txt
python $R/tools/pmc_sq.py /tmp/pmc2 igemm_x3 wgrad9_x3 wgrad_ups_x3 >> $R/$OUT.txt
python $R/tools/pmc_sq.py /tmp/pmc3 igemm_x3 wgrad9_x3 wgrad_ups_x3 >> $R/$OUT.txt
python - <<PY >> $R/$OUT.txt
import csv, glob
# kernel-trace durations of the same dispatches (pass 1) -> effective clock = GRBM_GUI_ACTIVE / duration
rows = {}
for f in glob.glob("/tmp/pmc1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "x3" in r["Kernel_Name"]:
            rows.setdefault(r["Kernel_Name"][:60], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in rows.items():
    print(f"== duration us {k}: n={len(v)} avg={sum(v)/len(v):.1f}")
PY
tail -3 /tmp/pmc1.log >> $R/$OUT.txt
