#!/bin/bash
# A/B of the persistent kernel's block size (256 x 3 per CU vs 768 x 1 per CU): full C4 frame, 1/2, 1/4 and 1/8 shares, C5.
# build first: (cd pearray_amd/csrc && make && make PP_BLOCK=768 B=build768 LIB=libprgpu768.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in libprgpu.so libprgpu768.so; do
  echo "== $lib"
  export PRGPU_LIBRARY=$R/pearray_amd/csrc/$lib
  for w in 1 2 4; do TILE=64 timeout -k 10 120 python $R/tools/gpu_probe_share8.py $w 96 | cut -c1-150 || exit 1; done
  TILE=16 timeout -k 10 120 python $R/tools/gpu_probe_share8.py 8 96 | cut -c1-150 || exit 1
  timeout -k 10 200 python $R/tools/gpu_c5.py 2>&1 | tail -2 || exit 1
done
