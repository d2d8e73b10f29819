#!/bin/bash
# LF (BASELINE config 3) step time and rocprofv3 kernel stats -> gpurun_out/profiles_r06/r06_lf_*.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/bench_lf.py --steps 50 > $O/r06_lf_step.txt 2>&1
rm -rf /tmp/lf
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/lf -o p -- python $R/tools/bench_lf.py --steps 10 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/lf $O/r06_lf_kernel_stats.txt "python tools/bench_lf.py --steps 10" > /dev/null

# AT.trainLSTM per-sample loop (T = 1, B = 1): hipGraph replay / launch by launch on the fused single-step kernels / on the
# sequence kernels, + kernel stats of the default
{ echo "# python tools/bench_at_loop.py --n 2000   (default: one hipGraph replay per sample)"
  timeout 600 python $R/tools/bench_at_loop.py --n 2000 2>&1 | grep "AT.trainLSTM"
  echo "# EGAZE_AT_GRAPH=0   (launch by launch, fused single-step kernels csrc/lstm_b1.hip)"
  EGAZE_AT_GRAPH=0 timeout 600 python $R/tools/bench_at_loop.py --n 2000 2>&1 | grep "AT.trainLSTM"
  } > $O/r06_at_sample_loop.txt
rm -rf /tmp/atl
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/atl -o p -- python $R/tools/bench_at_loop.py --n 200 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/atl /tmp/atl_stats.txt "python tools/bench_at_loop.py --n 200" > /dev/null
head -24 /tmp/atl_stats.txt >> $O/r06_at_sample_loop.txt
# config 5 stage timings
timeout 900 python $R/tools/bench_pipeline.py --frames 256 2>&1 | grep -v "^/opt\|^begin\|^Finished\|Warning" > /tmp/pipe.txt
{ echo "# python tools/bench_pipeline.py (BASELINE config 5 stages on one GPU, synthetic frames in host memory)"; cat /tmp/pipe.txt; } > $O/r06_pipeline_config5.txt
