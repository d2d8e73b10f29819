#!/bin/bash
# A/B of kernel build variants (csrc/build.sh with EGZ_VARIANT=...) on the conv microbenchmark.
# Usage: bash tools/ab_conv.sh "<variants...>" <bench_conv args...>
VARS=$1; shift
V=egocentric-gaze-prediction_amd/csrc/variants
echo "=== base"; python tools/bench_conv.py "$@" 2>&1 | grep -v amdgpu.ids
for v in $VARS; do
  echo "=== $v"; EGAZE_HIP_LIB=$PWD/$V/libegaze_hip_$v.so python tools/bench_conv.py "$@" 2>&1 | grep -v amdgpu.ids
done
