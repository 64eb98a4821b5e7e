run() { timeout 400 python bench.py --steps 300 --warmup 50 --repeats 1 --stamp-bursts 0 --no-cpu-baseline --no-secondary --no-pmc --trace-steps 10 "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = j['population']
print('$*', p['learners'], p['groups'], p['aggregate_updates_per_s'], p['ms_per_replay'], p['roofline'].get('fp32_frac'))" | tee -a gpurun_out/r5pop5/groups.txt; }
mkdir -p gpurun_out/r5pop5
run --population-learners 128 --population-groups 2
run --population-learners 128 --population-groups 4
run --population-learners 192 --population-groups 3
run --population-learners 256 --population-groups 4
run --population-learners 128 --population-groups 2
