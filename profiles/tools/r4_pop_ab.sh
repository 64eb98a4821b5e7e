#!/bin/bash
# Interleaved A/B of a population-path switch on ONE box: bash profiles/tools/r4_pop_ab.sh <tag> <ENV=val> [rounds] [extra bench args]
TAG=$1; SW=$2; N=${3:-3}; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { env "$1" timeout 400 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-secondary --population-wide 0 --trace-steps 10 "${@:2}" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = j['population']
print('$1', p['groups'], p['aggregate_updates_per_s'], p['ms_per_replay'], p['roofline'].get('fp32_frac'))" | tee -a $OUT/ab.txt; }
for i in $(seq $N); do run A=base "$@"; run $SW "$@"; done
