#!/bin/bash
# Round 6 GPU passes (run on the GPU box from the repo root):  IL_COMMIT=<sha> bash profiles/tools/r6_gpu_pass.sh <tag> what...
#   ab_envs    interleaved A/B of several environment settings on the headline line: IL_AB_SETS="A=1,B=2 A=0 ..." (space-separated sets, comma-separated VAR=value pairs)
#   tests      pytest -m gpu (the whole suite)                                          -> pytest_gpu.log
#   bench      python bench.py (the default line: 5 repeats, stamps, population, secondary, cpu_baseline) -> bench.json
#   driver     python bench.py --gpus 1 --steps 20 --warmup 5 (what the driver runs; rate only)            -> bench_driver.json
#   headline   rocprofv3 --kernel-trace of the timed schedule                           -> headline_kernel_stats.md (+ bench line of that run: stamps vs trace)
#   pmc        FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-graph --no-overlap`      -> pmc_latest.json, pmc.md
#   dp         one rank through the data-parallel schedules: peer fused / peer launches / RCCL all-reduces in the graph / plain plan -> dp_one_rank.txt (+ rccl kernel trace)
#   acting     profiles/tools/acting_bench.py                                            -> acting.json
#   gmmil pwil pop32   kernel traces of the secondary workloads
#   sqpop      SQ / TCP counters + FETCH/WRITE of the 32-learner population launches    -> sq_population.md
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-population --no-secondary"
line() { python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j['roofline']
print('$1', j['value'], j['ms_per_step'], j.get('timed_replays'), r.get('kernel'), r.get('frac'), {k: (v['avg_us'], v.get('active_us')) for k, v in r['kernels'].items()}, r['update'].get('launch_boundaries_us'), j['config'].get('exchange'), 'finite', j['config'].get('finite'))"; }
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
trace() {   # trace <name> <cmd>
  rm -rf /tmp/prof_$1
  (cd $ROOT && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$1 -o $1 -- $2 > $OUT/$1.log 2>&1)
  db=$(find /tmp/prof_$1 -name "*.db" | head -1)
  if [ -n "$db" ]; then python profiles/summarize_rocpd.py $db > $OUT/${1}_kernel_stats.md; else echo "no db for $1" > $OUT/${1}_kernel_stats.md; tail -n 5 $OUT/$1.log >> $OUT/${1}_kernel_stats.md; fi
}
for what in "$@"; do
  echo "== $what $(date +%T)" | tee -a $OUT/summary.txt
  case $what in
    tests)
      IL_FRACTIONS_OUT=$OUT/fractions.json timeout 1500 python -m pytest tests/ -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $OUT/summary.txt; tail -n 4 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt ;;
    tests_noexit)
      IL_FRACTIONS_OUT=$OUT/fractions.json timeout 1500 python -m pytest tests/ -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $OUT/summary.txt; tail -n 25 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt ;;
    bench)
      timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt; line full < $OUT/bench.json | tee -a $OUT/summary.txt ;;
    driver)
      for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $QUICK 2>/dev/null | tee $OUT/bench_driver_$i.json | line driver$i | tee -a $OUT/summary.txt; done ;;
    headline)
      trace headline "python bench.py --steps 400 --warmup 50 --repeats 3 --min-seconds 0 --stamp-bursts 10 $QUICK"
      grep -h '"metric"' $OUT/headline.log | tail -n 1 > $OUT/headline_bench.json; line traced < $OUT/headline_bench.json | tee -a $OUT/summary.txt
      head -n 30 $OUT/headline_kernel_stats.md | tee -a $OUT/summary.txt ;;
    pmc)
      HEAD="python bench.py --steps 100 --warmup 10 --repeats 1 --min-seconds 0 --stamp-bursts 0 --no-graph --no-overlap --no-pmc $QUICK --trace-steps 2"
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$c
        (cd $ROOT && timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_$c -o p -- $HEAD > $OUT/pmc_$c.log 2>&1)
      done
      f=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
      python profiles/tools/make_pmc_json.py $f $w "$HEAD" > $OUT/pmc_latest.json 2> $OUT/pmc_json.err
      python profiles/pmc_summary.py $f $w > $OUT/pmc.md 2>> $OUT/pmc_json.err
      head -n 20 $OUT/pmc.md | tee -a $OUT/summary.txt ;;
    dp)
      # one rank pushing to its own window / all-reducing with itself: what the schedules cost before any fabric is involved. RCCL: backend nccl, world 1, captured in the graph.
      runq() { env "$@" timeout 400 python bench.py --steps 2000 --warmup 200 $QUICK --stamp-bursts 5 2> $OUT/dp_last.err | line "$*" | tee -a $OUT/dp_one_rank.txt; }
      runq A=plain_single_gpu_plan
      runq IL_FORCE_DP=1 IL_PEER_EXCHANGE=force
      runq IL_FORCE_DP=1 IL_PEER_EXCHANGE=force IL_DP_FUSED=0
      runq IL_FORCE_DP=1 IL_FORCE_ALLREDUCE=1 IL_PEER_EXCHANGE=0 IL_BENCH_BACKEND=nccl
      runq IL_FORCE_DP=1 IL_FORCE_ALLREDUCE=1 IL_PEER_EXCHANGE=0 IL_BENCH_BACKEND=nccl IL_DP_HANDOFF=0
      trace dp_rccl "env IL_FORCE_DP=1 IL_FORCE_ALLREDUCE=1 IL_PEER_EXCHANGE=0 IL_BENCH_BACKEND=nccl python bench.py --steps 400 --warmup 50 --repeats 1 --min-seconds 0 $QUICK --stamp-bursts 0"
      cat $OUT/dp_one_rank.txt >> $OUT/summary.txt; head -n 24 $OUT/dp_rccl_kernel_stats.md | tee -a $OUT/summary.txt ;;
    acting)
      timeout 600 python profiles/tools/acting_bench.py > $OUT/acting.json 2> $OUT/acting.err; echo "acting rc=$?" | tee -a $OUT/summary.txt; tail -n 3 $OUT/acting.json | cut -c1-1500 | tee -a $OUT/summary.txt ;;
    ab_direct)
      # interleaved on one box: hipGraph replays against direct launches of the same two branches (UpdatePlan.launch_direct)
      for i in 1 2 3; do for e in ${IL_AB_EARLY:-1}; do for m in graph direct; do IL_EARLY_DRAW=$e timeout 300 python bench.py --launch $m --steps 2000 --warmup 200 --repeats 3 --stamp-bursts 10 $QUICK 2>$OUT/ab_direct.err | line "early_draw=$e launch=$m" | tee -a $OUT/ab_direct.txt; done; done; done
      cat $OUT/ab_direct.txt >> $OUT/summary.txt ;;
    ab_env)
      # interleaved A/B of one environment switch on the headline line: IL_AB_SWITCH=NAME (values 1 / 0), direct launches
      for i in 1 2 3; do for v in 1 0; do env $IL_AB_SWITCH=$v timeout 300 python bench.py --steps 2000 --warmup 200 --repeats 3 --stamp-bursts 10 $QUICK 2>$OUT/ab_env.err | line "$IL_AB_SWITCH=$v" | tee -a $OUT/ab_$IL_AB_SWITCH.txt; done; done
      cat $OUT/ab_$IL_AB_SWITCH.txt >> $OUT/summary.txt ;;
    ab_envs)
      for i in 1 2 3; do for set in $IL_AB_SETS; do env $(echo $set | tr ',' ' ') timeout 300 python bench.py --steps 2000 --warmup 200 --repeats 3 --stamp-bursts 10 $QUICK 2>$OUT/ab_envs.err | line "$set" | tee -a $OUT/ab_envs.txt; tail -n 2 $OUT/ab_envs.err | grep -v "^$" | cut -c1-300 >> $OUT/ab_envs.txt; done; done
      cat $OUT/ab_envs.txt >> $OUT/summary.txt ;;
    ab_gmmil_rb8)
      for i in 1 2 3; do for lib in "" "$ROOT/variants/rb8/libil_hip.so"; do IL_HIP_LIBRARY=$lib timeout 300 python profiles/tools/secondary_workloads.py gmmil_rate 2>$OUT/ab_gmmil.err | tail -n 1 | sed "s|^|lib=${lib:-in-tree} |" | tee -a $OUT/ab_gmmil_rb8.txt; done; done
      cat $OUT/ab_gmmil_rb8.txt >> $OUT/summary.txt
      IL_HIP_LIBRARY=$ROOT/variants/rb8/libil_hip.so trace gmmil_rb8 "python profiles/tools/secondary_workloads.py gmmil"; head -n 4 $OUT/gmmil_rb8_kernel_stats.md | tee -a $OUT/summary.txt ;;
    popline)
      timeout 600 python bench.py --steps 300 --warmup 50 --repeats 1 --stamp-bursts 0 --no-cpu-baseline --no-secondary --trace-steps 10 2>$OUT/popline.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = j['population']
print('population', p['learners'], p['groups'], p['aggregate_updates_per_s'], p['ms_per_replay'], p['roofline'].get('fp32_frac'))" | tee -a $OUT/summary.txt ;;
    ab_gmmil_mfma)
      # round 6: the centred Gram form on the matrix pipes (k_gmmil_mfma, default) against the direct-difference launch (k_gmmil_sx), interleaved on one box
      for i in 1 2 3; do for m in 1 0; do IL_GMMIL_MFMA=$m timeout 300 python profiles/tools/secondary_workloads.py gmmil_rate 2>$OUT/ab_gmmil.err | tail -n 1 | sed "s/^/IL_GMMIL_MFMA=$m /" | tee -a $OUT/ab_gmmil_mfma.txt; done; done
      for m in 1 0; do IL_GMMIL_MFMA=$m timeout 300 python profiles/tools/gmmil_ab.py 2>>$OUT/ab_gmmil.err | tail -n 1 | sed "s/^/IL_GMMIL_MFMA=$m /" | tee -a $OUT/ab_gmmil_mfma.txt; done
      cat $OUT/ab_gmmil_mfma.txt >> $OUT/summary.txt ;;
    ab_gmmil_sx)
      for i in 1 2 3; do for m in 1 0; do IL_GMMIL_SX=$m timeout 300 python profiles/tools/secondary_workloads.py gmmil_rate 2>$OUT/ab_gmmil.err | tail -n 1 | sed "s/^/IL_GMMIL_SX=$m /" | tee -a $OUT/ab_gmmil_sx.txt; done; done
      cat $OUT/ab_gmmil_sx.txt >> $OUT/summary.txt ;;
    ab_gmmil)
      for i in 1 2 3; do for m in 1 0; do IL_GMMIL_RESIDENT=$m timeout 300 python profiles/tools/secondary_workloads.py gmmil_rate 2>$OUT/ab_gmmil.err | tail -n 1 | sed "s/^/IL_GMMIL_RESIDENT=$m /" | tee -a $OUT/ab_gmmil.txt; done; done
      cat $OUT/ab_gmmil.txt >> $OUT/summary.txt ;;
    tests_k)
      IL_FRACTIONS_OUT=$OUT/fractions_k.json timeout 1500 python -m pytest tests/ -m gpu -q -k "$IL_TESTS_K" > $OUT/pytest_k.log 2>&1; echo "pytest -k rc=$?" | tee -a $OUT/summary.txt; tail -n 12 $OUT/pytest_k.log | tee -a $OUT/summary.txt ;;
    gmmil_plan)
      rm -rf /tmp/prof_gp; (cd $ROOT && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_gp -o gp -- python profiles/tools/gmmil_plan_trace.py > $OUT/gmmil_plan.log 2>&1)
      db=$(find /tmp/prof_gp -name "*.db" | head -1); [ -n "$db" ] && python profiles/summarize_rocpd.py $db > $OUT/gmmil_plan_kernel_stats.md; tail -n 1 $OUT/gmmil_plan.log | tee -a $OUT/summary.txt; head -n 14 $OUT/gmmil_plan_kernel_stats.md | tee -a $OUT/summary.txt ;;
    gmmil) trace gmmil "python profiles/tools/secondary_workloads.py gmmil"; head -n 12 $OUT/gmmil_kernel_stats.md | tee -a $OUT/summary.txt ;;
    pwil) trace pwil "python profiles/tools/secondary_workloads.py pwil"; head -n 12 $OUT/pwil_kernel_stats.md | tee -a $OUT/summary.txt ;;
    pop32) trace pop32 "python profiles/tools/secondary_workloads.py population 32"; head -n 24 $OUT/pop32_kernel_stats.md | tee -a $OUT/summary.txt ;;
    sqpop) counters sq_population "python profiles/tools/secondary_workloads.py population 32" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY FETCH_SIZE WRITE_SIZE ;;
    sqhead) counters sq_headline "python bench.py --steps 100 --warmup 10 --repeats 1 --min-seconds 0 --stamp-bursts 0 --no-graph --no-overlap --no-pmc $QUICK --trace-steps 2" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY ;;
    *) echo "unknown pass $what" | tee -a $OUT/summary.txt ;;
  esac
done
echo "== done $(date +%T)" | tee -a $OUT/summary.txt
