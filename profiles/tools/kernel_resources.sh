#!/bin/bash
# VGPRs / scratch / spills / occupancy of every kernel of one .hip file (hipcc remarks; no GPU needed): bash profiles/tools/kernel_resources.sh sac.hip [extra hipcc flags]
# Used before / after a change to the shared device headers: the 16-wave tile kernels sit just under the 128-VGPR budget, and a cold path that pushes one over
# costs more than any of the optimisations in DESIGN.md gained.
cd "$(dirname "$0")/../../imitation-learning_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 \
  | grep -E "Function Name|    VGPRs:|ScratchSize|VGPRs Spill|Occupancy|LDS Size" | sed -E 's/^[^ ]+ remark: +//; s/ \[-Rpass.*//' | paste - - - - - - \
  | awk -F'\t' '{gsub("Function Name: ","",$1); printf "%-70s %s | %s | %s | %s | %s\n", $1, $2, $3, $4, $5, $6}'
