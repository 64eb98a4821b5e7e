for i in 1 2 3; do
  for v in pf nopf; do
    if [ $v = nopf ]; then export IL_HIP_LIBRARY=/root/repo/imitation-learning_amd/ab/libil_hip_nopf.so; else unset IL_HIP_LIBRARY; fi
    python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-population --trace-steps 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'])"
  done
done
