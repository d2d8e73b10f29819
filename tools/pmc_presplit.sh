#!/bin/bash
# SQ counters of the forward / weight-gradient kernels with fp32 and with pre-split operands (VALU instructions per MFMA before / after).
# Usage (GPU box): bash tools/pmc_presplit.sh <out file> <shape substring e.g. enc10>
OUT=$1; SH=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
rm -rf /tmp/pps
timeout 280 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pps -o p -- python $R/tools/bench_conv.py --dtype 1 --only $SH --iters 5 --presplit --what fwd,wgrad > /tmp/pps.log 2>&1 || tail -5 /tmp/pps.log
{ echo "# rocprofv3 --pmc $P -- python tools/bench_conv.py --dtype 1 --only $SH --iters 5 --presplit --what fwd,wgrad"
  echo "# per launch; igemm ...Li0ELb0EE = forward over fp32 activations (split at staging), ...Li0ELb1EE = over pre-split pairs;"
  echo "# wgrad9 ...ELb0ELb0EE = fp32 x and dy, ...ELb1ELb0EE = pre-split x"
  python $R/tools/pmc_sq.py /tmp/pps "x3s_kernelIDF16_Li1ELi2ELb1ELi0ELb0EE" "x3s_kernelIDF16_Li1ELi2ELb1ELi0ELb1EE" "x3s_kernelIDF16_Li2ELi2ELb1ELi0ELb0EE" "x3s_kernelIDF16_Li2ELi2ELb1ELi0ELb1EE" "x3s_kernelIDF16_Li1ELi2ELb0ELi0ELb0EE" "x3s_kernelIDF16_Li1ELi2ELb0ELi0ELb1EE" "wgrad9_x3_kernelIDF16_Lb0ELi4ELi8ELb0ELb0E" "wgrad9_x3_kernelIDF16_Lb0ELi4ELi8ELb1ELb0E"
} > $R/$OUT
