#!/bin/bash
# Round 4, GPU pass of the pair-mode kernels (run on the GPU box from the repo root): parity, same-box A/B against IL_PAIR=0, both timelines, kernel traces.
#   bash profiles/tools/r4_pair_ab.sh <tag> [rounds] [notests]
TAG=${1:-r04a}; N=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "$3" != notests ]; then
timeout 900 python -m pytest tests/test_timed_path_oracle.py tests/test_update_plans_gpu.py -m gpu -x -q > $OUT/pytest_timed.log 2>&1; echo "pytest timed rc=$?" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sac or schedule_switches or handoff or plan or relabel or pair_mode" > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$?" | tee -a $OUT/summary.txt
for f in $OUT/pytest_timed.log $OUT/pytest_parity.log; do tail -n 3 $f | tee -a $OUT/summary.txt; done
fi
run() { env "$@" timeout 300 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-population --no-secondary --trace-steps 50 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', j['value'], j['ms_per_step'], {k: v['avg_us'] for k, v in j['roofline']['kernels'].items()})" | tee -a $OUT/ab.txt; }
for i in $(seq $N); do
  run IL_PAIR=1; run IL_PAIR=0
  run IL_PAIR=1 IL_HIP_LIBRARY=$PWD/variants/tl/libil_hip.so; run IL_PAIR=0 IL_HIP_LIBRARY=$PWD/variants/tl/libil_hip.so
done
IL_HIP_LIBRARY=$PWD/variants/tl/libil_hip.so IL_PAIR=1 timeout 300 python profiles/tools/pair_timeline.py 50 > $OUT/timeline_pair.txt 2>&1
IL_HIP_LIBRARY=$PWD/variants/tl/libil_hip.so IL_PAIR=0 timeout 300 python profiles/tools/update_timeline.py 50 > $OUT/timeline_nopair.txt 2>&1
for v in 1 0; do
  rm -rf /tmp/prof_$v
  IL_PAIR=$v rocprofv3 --kernel-trace -d /tmp/prof_$v -o t -- python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-population --no-secondary --trace-steps 2 > $OUT/trace_$v.log 2>&1
  db=$(find /tmp/prof_$v -name "*.db" | head -1)
  [ -n "$db" ] && python profiles/summarize_rocpd.py $db > $OUT/kernel_stats_pair$v.md && cp $db $OUT/trace_pair$v.db
done
tail -n 5 $OUT/timeline_pair.txt
