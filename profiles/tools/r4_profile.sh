#!/bin/bash
# Round 4 profiling passes (run on the GPU box from the repo root): bash profiles/tools/r4_profile.sh <tag> what...
#   headline   rocprofv3 --kernel-trace of the timed schedule (two graphs, pair mode)          -> <what>_kernel_stats.md
#   pmc        FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-graph --no-overlap`              -> pmc_latest.json, pmc.md
#   sqhead     SQ / GRBM counters of the same run, one pass per counter                         -> sq_headline.md
#   sqpop      SQ / TCP / GRBM counters of the 32-learner population launches                   -> sq_population.md  (which resource binds: VERDICT r3 #7)
#   gmmil pwil pop32 kernel traces of the secondary workloads
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
HEAD="python bench.py --steps 100 --warmup 10 --no-graph --no-overlap --no-cpu-baseline --no-population --no-secondary --trace-steps 2"
POP="python profiles/tools/secondary_workloads.py population 32"
counters() {   # counters <name> <cmd> <counter>...
  local name=$1 cmd=$2; shift 2
  : > $OUT/$name.md
  for c in "$@"; do
    rm -rf /tmp/sq_$c
    (cd $ROOT && timeout 600 rocprofv3 --pmc $c -d /tmp/sq_$c -o p -- $cmd > $OUT/${name}_$c.log 2>&1)
    db=$(find /tmp/sq_$c -name "*.db" | head -1)
    [ -n "$db" ] && python profiles/pmc_summary.py $db >> $OUT/$name.md || echo "no db for $c" >> $OUT/$name.md
  done
}
for what in "$@"; do
  case $what in
    pwil) cmd="python profiles/tools/secondary_workloads.py pwil" ;;
    gmmil) cmd="python profiles/tools/secondary_workloads.py gmmil" ;;
    pop32) cmd="$POP" ;;
    headline) cmd="python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-population --no-secondary --trace-steps 2" ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$c
        (cd $ROOT && timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_$c -o p -- $HEAD > $OUT/pmc_$c.log 2>&1)
      done
      f=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
      python profiles/tools/make_pmc_json.py $f $w > $OUT/pmc_latest.json 2> $OUT/pmc_json.err
      python profiles/pmc_summary.py $f $w > $OUT/pmc.md 2>> $OUT/pmc_json.err
      continue ;;
    sqhead) counters sq_headline "$HEAD" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; continue ;;
    sqpop) counters sq_population "$POP" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY TCP_PENDING_STALL_CYCLES SQ_LDS_BANK_CONFLICT FETCH_SIZE WRITE_SIZE; continue ;;
  esac
  rm -rf /tmp/prof_$what
  (cd $ROOT && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$what -o $what -- $cmd > $OUT/$what.log 2>&1)
  db=$(find /tmp/prof_$what -name "*.db" | head -1)
  if [ -n "$db" ]; then python profiles/summarize_rocpd.py $db > $OUT/${what}_kernel_stats.md; else echo "no db for $what" > $OUT/${what}_kernel_stats.md; tail -n 5 $OUT/$what.log >> $OUT/${what}_kernel_stats.md; fi
done
