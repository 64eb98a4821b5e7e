#!/bin/bash
# Interleaved A/B of the population line between the in-tree library and variants/<name>/libil_hip.so builds on ONE box:
#   bash profiles/tools/r5_pop_ab.sh <tag> <rounds> <variant> [variant ...]        (variants: profiles/tools/build_variants.sh)
TAG=$1; N=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { env "$2" timeout 400 python bench.py --steps 300 --warmup 50 --repeats 1 --stamp-bursts 0 --no-cpu-baseline --no-secondary --no-pmc --trace-steps 10 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = j['population']
print('$1', p['learners'], p['groups'], p['aggregate_updates_per_s'], p['ms_per_replay'], p['roofline'].get('fp32_frac'))" | tee -a $OUT/ab.txt; }
for i in $(seq $N); do
  run in-tree A=1
  for v in "$@"; do run $v IL_HIP_LIBRARY=$PWD/variants/$v/libil_hip.so; done
done
