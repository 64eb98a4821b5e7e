#!/bin/bash
# rocprofv3 --kernel-trace passes of round 3 (run on the GPU box from the repo root): bash profiles/tools/r3_profile.sh <tag> what...   what in: pwil gmmil pop32 headline dp
# Summaries land in gpurun_out/<tag>/<what>_kernel_stats.md (copy the ones to keep into profiles/).
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    pwil) cmd="python profiles/tools/secondary_workloads.py pwil" ;;
    gmmil) cmd="python profiles/tools/secondary_workloads.py gmmil" ;;
    pop32) cmd="python profiles/tools/secondary_workloads.py population 32" ;;
    headline) cmd="python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-population --no-secondary --trace-steps 2" ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$c
        (cd $ROOT && rocprofv3 --pmc $c -d /tmp/pmc_$c -o p -- python bench.py --steps 100 --warmup 10 --no-graph --no-overlap --no-cpu-baseline --no-population --no-secondary --trace-steps 2 > $OUT/pmc_$c.log 2>&1)
      done
      f=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
      python profiles/tools/make_pmc_json.py $f $w > $OUT/pmc_latest.json 2> $OUT/pmc_json.err
      python profiles/pmc_summary.py $f $w > $OUT/pmc.md 2>> $OUT/pmc_json.err
      continue ;;
    pop32pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/ppmc_$c
        (cd $ROOT && rocprofv3 --pmc $c -d /tmp/ppmc_$c -o p -- python profiles/tools/secondary_workloads.py population 32 > $OUT/pop32pmc_$c.log 2>&1)
      done
      f=$(find /tmp/ppmc_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/ppmc_WRITE_SIZE -name "*.db" | head -1)
      python profiles/tools/make_pmc_json.py $f $w > $OUT/pop32_pmc.json 2> $OUT/pop32_pmc.err
      continue ;;
    dp) cmd="env IL_FORCE_DP=1 IL_PEER_EXCHANGE=force python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-population --no-secondary --trace-steps 2" ;;
  esac
  rm -rf /tmp/prof_$what
  (cd $ROOT && rocprofv3 --kernel-trace -d /tmp/prof_$what -o $what -- $cmd > $OUT/$what.log 2>&1)
  db=$(find /tmp/prof_$what -name "*.db" | head -1)
  if [ -n "$db" ]; then python profiles/summarize_rocpd.py $db > $OUT/${what}_kernel_stats.md; cp $db $OUT/${what}_results.db 2>/dev/null; else echo "no db for $what" > $OUT/${what}_kernel_stats.md; tail -5 $OUT/$what.log >> $OUT/${what}_kernel_stats.md; fi
done
