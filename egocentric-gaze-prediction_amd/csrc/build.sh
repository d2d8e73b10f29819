#!/bin/bash
# Builds libegaze_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
SRCS=$(ls *.hip)
OUT=libegaze_hip.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wno-unused-result -Wl,-z,defs "$@" $SRCS -o $OUT.tmp
mv $OUT.tmp $OUT
echo "built $(pwd)/$OUT"
