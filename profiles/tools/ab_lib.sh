# A/B two builds of libil_hip.so on ONE box (box-to-box variance is ~10%): bash profiles/tools/ab_lib.sh <alt .so> [rounds]
ALT=$1; N=${2:-3}
for i in $(seq $N); do
  for v in new alt; do
    if [ $v = alt ]; then export IL_HIP_LIBRARY=$ALT; else unset IL_HIP_LIBRARY; fi
    python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-population --trace-steps 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'])"
  done
done
