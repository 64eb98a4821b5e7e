#!/bin/bash
# Interleaved A/B of the variants built by build_variants.sh on ONE box (box-to-box variance is ~8 %): bash profiles/tools/ab_variants.sh <rounds> <name> [<name> ...]
# ("default" = the in-tree library). One line per run: name, updates/s, ms per update, eager per-kernel HIP-event averages.
N=$1; shift
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT"
for i in $(seq $N); do
  for v in "$@"; do
    unset IL_DW_BLOCK32
    if [ "$v" = default ]; then unset IL_HIP_LIBRARY; elif [ "$v" = noblock32 ]; then unset IL_HIP_LIBRARY; export IL_DW_BLOCK32=0; else export IL_HIP_LIBRARY="$ROOT/variants/$v/libil_hip.so"; fi
    python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-population --no-secondary --trace-steps 50 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', j['value'], j['ms_per_step'], {k: v['avg_us'] for k, v in j['roofline']['kernels'].items()})"
  done
done
