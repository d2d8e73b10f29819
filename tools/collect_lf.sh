#!/bin/bash
# LF (BASELINE config 3) step time and rocprofv3 kernel stats -> gpurun_out/profiles_r02/r02_lf_*.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_lf.py --steps 50 > $O/r02_lf_step.txt 2>&1
rm -rf /tmp/lf
rocprofv3 --kernel-trace --output-format csv -d /tmp/lf -o p -- python $R/tools/bench_lf.py --steps 10 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/lf $O/r02_lf_kernel_stats.txt "python tools/bench_lf.py --steps 10" > /dev/null
