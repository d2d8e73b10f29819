#!/bin/bash
# Register / LDS / spill report of the kernels of one .hip source (gfx950): tools/kres.sh csrc-file [grep pattern] [extra flags]
# Usage: bash tools/kres.sh egocentric-gaze-prediction_amd/csrc/conv3x3_igemm_x3s.hip x3s_kernel
SRC=$1; PAT=${2:-.}; shift; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -I"$(dirname $SRC)" -c $SRC -o /tmp/kres.o \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r"Function Name: (\S+)",ln)
    if m:
        cur={"name":m.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for key in ("VGPRs","AGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","SGPRs","LDS Size [bytes/block]","VGPR Spill"):
        m=re.search(re.escape(key)+r": (\d+)",ln)
        if m: cur[key]=int(m.group(1))
names=subprocess.run(["/usr/bin/c++filt"]+[r["name"] for r in rows],capture_output=True,text=True).stdout.split("\n")
for r,n in zip(rows,names):
    n=re.sub(r"\(anonymous namespace\)::","",n); n=re.sub(r"\(.*","",n)
    m=re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)I(DF16_|DF16b)((?:L[ib]\d+E)+)E",r["name"])
    if m: n=m.group(1)+"<"+("f16" if m.group(2)=="DF16_" else "bf16")+","+",".join(re.findall(r"L[ib](\d+)E",m.group(3)))+">"
    print("%-90s v=%3d a=%3d scratch=%4d occ=%d lds=%6d" % (n[:90], r.get("VGPRs",-1), r.get("AGPRs",-1), r.get("ScratchSize [bytes/lane]",-1), r.get("Occupancy [waves/SIMD]",-1), r.get("LDS Size [bytes/block]",-1)))
' | grep "$PAT"
