#!/bin/bash
# Builds A/B variants of libil_hip.so (no GPU needed) into variants/<name>/libil_hip.so:  bash profiles/tools/build_variants.sh name1:"-Dflags" name2:"-Dflags" ...
# variants/ is git-ignored (*.so, *.o) but travels to the GPU box; profiles/tools/ab_variants.sh runs them interleaved on one box (IL_HIP_LIBRARY).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"; [ "$flags" = "$spec" ] && flags=""
  mkdir -p "$ROOT/variants/$name/o"
  make -s -C "$ROOT/imitation-learning_amd/csrc" -j8 ARCH=gfx950 EXTRA="$flags" BUILD="$ROOT/variants/$name/o" OUT="$ROOT/variants/$name/libil_hip.so" >/dev/null
  echo "$flags" > "$ROOT/variants/$name/flags.txt"
  echo "built variants/$name ($flags)"
done
