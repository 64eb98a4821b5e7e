#!/bin/bash
# Developer tool: what the shader clock does while the population line runs - rocm-smi polled every 0.5 s beside `bench.py` (population leg): bash profiles/tools/pop_clock_poll.sh <out dir>
OUT=$1; mkdir -p $OUT
( timeout 300 python bench.py --steps 300 --warmup 50 --repeats 1 --stamp-bursts 0 --no-cpu-baseline --no-secondary --no-pmc --trace-steps 10 > $OUT/clock_poll_bench.json 2>/dev/null ) &
BP=$!
: > $OUT/clock_poll.txt
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> $OUT/clock_poll.txt
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use|busy" >> $OUT/clock_poll.txt
  sleep 0.5
done
wait $BP
