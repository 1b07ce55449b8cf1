#!/bin/bash
# Tile shares against the number of resident blocks per CU (fewer waves per SIMD = a faster pace for every single path, less throughput):
#   bash tools/gpu_share_blocks.sh TAG c4|c5 "WORLDS" -> gpurun_out/TAG.log
TAG=${1:-share_blocks}; WL=${2:-c5}; WORLDS=${3:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"; : > "$OUT/$TAG.log"
for B in 0 2 1; do
  for S in 384 512; do
    PRGPU_PP_BLOCKS_PER_CU=$B PRGPU_PP_SLOTS=$S timeout -k 10 300 python3 $R/tools/gpu_shares.py 32 $WORLDS --workload $WL >> "$OUT/$TAG.log" 2>&1
  done
done
cat "$OUT/$TAG.log"
