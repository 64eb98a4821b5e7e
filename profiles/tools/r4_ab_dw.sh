OUT=gpurun_out/r04r; mkdir -p $OUT
run() { env "$@" timeout 300 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-population --no-secondary --trace-steps 50 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', j['value'], j['ms_per_step'])" | tee -a $OUT/ab.txt; }
for i in 1 2 3; do run A=new; run IL_HIP_LIBRARY=$PWD/variants/prevdw/libil_hip.so; done
IL_HIP_LIBRARY=$PWD/variants/tl/libil_hip.so python profiles/tools/dw_stragglers.py > $OUT/dw.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_timed_path_oracle.py tests/test_parallel_gpu.py -m gpu -q -x -k "sac or adam or pair_mode or oracle or parallel or peer or bc" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log
tail -n 3 $OUT/pytest.log
