#!/bin/bash
# A/B of two builds of the library (same ABI) on the share probes: usage: bash tools/gpu_ab_lib.sh libprgpu.so libprgpu_split.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  echo "== $lib"
  export PRGPU_LIBRARY=$R/pearray_amd/csrc/$lib
  for w in 1 4; do TILE=64 timeout -k 10 120 python $R/tools/gpu_probe_share8.py $w 96 | cut -c1-200 || exit 1; done
  for w in 8 16; do TILE=16 timeout -k 10 120 python $R/tools/gpu_probe_share8.py $w 96 | cut -c1-200 || exit 1; done
done
