#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/profiles_r06/ (copy to profiles/ afterwards).
# Usage: gpurun -- 'bash tools/collect_profiles.sh'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# every profiled step is the DEFAULT arithmetic (three products per MAC everywhere): the opt-in-arithmetic leg and the exact-f32 leg are off
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-bwd-leg"

# 2. rocprofv3 kernel stats, streams on / off (the roofline leg's configuration)
rm -rf /tmp/ks1 /tmp/ks0
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks1 -o p -- $BENCH > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/ks1 $O/r06_bench_b32_kernel_stats.txt "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-bwd-leg" > /dev/null
python $R/tools/trace_mfma.py /tmp/ks1 > $O/r06_exposed_time.txt 2>&1
python $R/tools/trace_windows.py /tmp/ks1 8 > $O/r06_step_windows.txt 2>&1
EGAZE_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks0 -o p -- $BENCH > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/ks0 $O/r06_bench_b32_kernel_stats_streams0.txt "EGAZE_STREAMS=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-leg --no-bwd-leg" > /dev/null

# 3. HBM traffic per launch (separate --pmc passes), conv fwd + dgrad family and wgrad family
rm -rf /tmp/pf /tmp/pw
EGAZE_STREAMS=0 timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o p -- $BENCH > /dev/null 2>&1
EGAZE_STREAMS=0 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o p -- $BENCH > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw igemm_x3 $O/r06_pmc_traffic_conv_fwd_dgrad.json
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw wgrad9_x3,wgrad_ups_x3 $O/r06_pmc_traffic_wgrad.json

# 3b. the default bench line, after the traffic files so that it can quote them (bench.py reads profiles/ and checks
#     the kernel-source hash stamped into them)
cp $O/r06_pmc_traffic_conv_fwd_dgrad.json $O/r06_pmc_traffic_wgrad.json $R/profiles/
cd /tmp
timeout 900 python $R/bench.py > $O/r06_bench_b32.json 2> $O/r06_bench_b32.stderr

# 4. SQ / GRBM counters of the conv kernels on three layer shapes
cd $R
timeout 900 bash tools/pmc_conv.sh gpurun_out/profiles_r06/r06_sq_counters_enc27_512x512_28 --dtype 1 --only enc27 --iters 5 --presplit
timeout 900 bash tools/pmc_conv.sh gpurun_out/profiles_r06/r06_sq_counters_enc10_128x128_112 --dtype 1 --only enc10 --iters 5 --presplit
timeout 900 bash tools/pmc_conv.sh gpurun_out/profiles_r06/r06_sq_counters_dec26_64x64_224 --dtype 1 --only dec26 --iters 5 --presplit

# 5. DVFS probe: the same kernels on all-zero operands, and the conv microbenchmark on random data
timeout 600 python tools/bench_conv.py --dtype 1 --iters 20 > $O/r06_conv_microbench.txt 2>&1
timeout 600 python tools/bench_conv.py --dtype 1 --iters 20 --zero > $O/r06_conv_microbench_zero_operands.txt 2>&1

# 6. CPU baseline thread sweep (oracle SP train step, batch 8)
for t in 8 16 32 64; do timeout 600 python tests/report_cpu_baseline_sweep.py $t 8; done > $O/r06_cpu_baseline_thread_sweep.txt 2>&1
nproc >> $O/r06_cpu_baseline_thread_sweep.txt; lscpu | grep "Model name" >> $O/r06_cpu_baseline_thread_sweep.txt
# 6b. the oracle step at the headline batch 32 beside the bounded batch 8 (once per round)
timeout 900 python tools/cpu_baseline_b32.py 16 > $O/r06_cpu_baseline_b32.txt 2>&1
# 11. backward convolutions on two / three MFMA products per MAC: the step, alternating, and the accuracy report
{ for rep in 1 2; do for v in 3 2; do
    echo "--- EGAZE_BWD_PRODUCTS=$v"
    EGAZE_BWD_PRODUCTS=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('ms/step %.3f  regions %s' % (d['ms_per_step'], [round(x,3) for x in d['extra']['timed_repeats']['ms_per_step']]))"
  done; done; } 2>/dev/null | grep -E '^--- |^ms/step' > $O/r06_bwd_products_step_ab.txt
{ echo "# python -m pytest tests/test_hip_ops.py tests/test_hip_model_sp.py -m gpu -s -k 'headline_geometry or two_product'"
  echo "# per convolution of the SP step at batch 32 (max-relative error vs torch-CPU fp32: forward, data gradient, weight gradient with three"
  echo "# products | with two products: max-relative and relative L2), then the whole model: two-product vs three-product gradients"
  timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model_sp.py -q -m gpu -s -k "headline_geometry or two_product or trajectory_32 or grads_vs_fp64 and not surveyed" 2>&1 | grep -E "two product|two-product|products|headline geometry|passed|failed"; } > $O/r06_two_products_report.txt
{ echo "# python tests/report_headline_grads.py --batch 8   (default arithmetic: three products)"
  timeout 900 python tests/report_headline_grads.py --batch 8 2>&1 | grep -v "amdgpu.ids\|tensors below"; } > $O/r06_headline_grads.txt
# 12. the AT recurrence: wavefront launches against the persistent launches, phase trace of the persistent kernels
{ timeout 300 python tools/bench_lstm_seq.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# EGAZE_HIP_LIB=.../variants/libegaze_hip_trace.so python tools/lstm_persist_trace.py   (build: EGZ_VARIANT=trace csrc/build.sh -DEGZ_PERSIST_TRACE)"
  [ -f egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_trace.so ] && EGAZE_HIP_LIB=egocentric-gaze-prediction_amd/csrc/variants/libegaze_hip_trace.so timeout 120 python tools/lstm_persist_trace.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# tools/bench_at_step.py, alternating EGAZE_LSTM_PERSIST = 1 / 0"
  for p in 1 0 1 0; do echo "EGAZE_LSTM_PERSIST=$p"; EGAZE_LSTM_PERSIST=$p timeout 200 python tools/bench_at_step.py 2>/dev/null; done; } > $O/r06_lstm_persistent.txt
ls -la $O

# 7. LF (config 3): step time with / without the device-side metric, and its kernel stats
bash tools/collect_lf.sh

# 8. AT step (config 4 shape) alone: time and kernel stats
cd /tmp
{ timeout 300 python $R/tools/bench_at_step.py --steps 200 2>&1 | grep "AT step"; } > $O/r06_at_step.txt
rm -rf /tmp/ats
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ats -o p -- python $R/tools/bench_at_step.py --steps 50 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/ats /tmp/ats_stats.txt "python tools/bench_at_step.py --steps 50" > /dev/null
head -26 /tmp/ats_stats.txt >> $O/r06_at_step.txt

# 9. data-parallel path at world size 1 (three legs, two rounds)
{ echo "# python tools/dp_world1.py --rounds 2 --steps 8   (ms per SP + AT step: no process group | RCCL group of one rank, nothing attached | dp.GradReducer on)"
  timeout 600 python $R/tools/dp_world1.py --rounds 2 --steps 8 2>&1 | grep "^plain"; } > $O/r06_dp_world1.txt

# 10. pre-split microbenchmark and the step with / without the pairs
cd $R
timeout 600 python tools/bench_conv.py --dtype 1 --iters 20 --presplit --what fwd,wgrad 2>&1 | grep -v amdgpu.ids > $O/r06_presplit_microbench.txt
# (the pre-split step A/B of round 5 used environment switches that are module constants now: profiles/r05_presplit_step_ab.txt stands)
ls -la $O
