OUT=gpurun_out/r04h; mkdir -p $OUT
run() { env "$@" timeout 300 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-population --no-secondary --trace-steps 50 $EXTRA 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$* $EXTRA', j['value'], j['ms_per_step'])" | tee -a $OUT/ab.txt; }
for i in 1 2; do
  EXTRA="" run IL_PAIR=1
  EXTRA="--no-graph" run IL_PAIR=1
done
python profiles/tools/host_bound_probe.py > $OUT/host_bound.txt 2>&1
tail -n 20 $OUT/host_bound.txt
