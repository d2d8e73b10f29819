#!/bin/bash
# Builds libegaze_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
# One object per .hip source, compiled in parallel and only when the source (or a header) is newer than its object;
# extra flags force a full rebuild (they may change code generation).
# EGZ_VARIANT=name builds variants/libegaze_hip_<name>.so in its own object directory (kernel A/B runs, loaded with
# EGAZE_HIP_LIB=...); variants/ is git-ignored and travels to the GPU box like the main .so.
set -e
cd "$(dirname "$0")"
OUT=libegaze_hip.so
OBJ=build
if [ -n "$EGZ_VARIANT" ]; then
    mkdir -p variants
    OUT=variants/libegaze_hip_$EGZ_VARIANT.so
    OBJ=build_$EGZ_VARIANT
fi
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result"
if [ ! -f $OBJ/.flags ] || [ "$(cat $OBJ/.flags)" != "$FLAGS $*" ]; then
    rm -f $OBJ/*.o
fi
echo "$FLAGS $*" > $OBJ/.flags
NEWEST_HDR=$(ls -t *.h | head -1)
TODO=""
for s in *.hip; do
    o=$OBJ/${s%.hip}.o
    if [ ! -f $o ] || [ $s -nt $o ] || [ $NEWEST_HDR -nt $o ]; then TODO="$TODO $s"; fi
done
if [ -n "$TODO" ]; then
    echo $TODO | tr ' ' '\n' | xargs -P "$(nproc)" -I{} bash -c \
        's={}; hipcc '"$FLAGS $*"' -c $s -o '"$OBJ"'/${s%.hip}.o.tmp && mv '"$OBJ"'/${s%.hip}.o.tmp '"$OBJ"'/${s%.hip}.o'
fi
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs $OBJ/*.o -o $OUT.tmp
mv $OUT.tmp $OUT
echo "built $(pwd)/$OUT"
