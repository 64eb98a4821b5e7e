#!/bin/bash
# Interleaved A/B of the in-tree library against variants/<name>/libil_hip.so on ONE box + the pair-mode timeline of the in-tree build: bash profiles/tools/r4_ab_lib.sh <tag> <variant> [rounds]
TAG=$1; V=$2; N=${3:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { env "$@" timeout 300 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-population --no-secondary --trace-steps 50 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', j['value'], j['ms_per_step'])" | tee -a $OUT/ab.txt; }
for i in $(seq $N); do run A=new; run IL_HIP_LIBRARY=$PWD/variants/$V/libil_hip.so; done
IL_HIP_LIBRARY=$PWD/variants/tl/libil_hip.so IL_PAIR=1 timeout 300 python profiles/tools/pair_timeline.py 50 > $OUT/timeline_pair.txt 2>&1
