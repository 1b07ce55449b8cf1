#!/bin/bash
# C5 with the 2-waves-per-SIMD build of the all-features kernel (256 VGPRs, 185 spilled instead of 786) against the default 3-waves one
R=${GRAFT_REPO_ROOT:-$(pwd)}
export C5_CHECK=0
run() { echo "== $*"; env "$@" timeout -k 10 200 python $R/tools/gpu_c5.py 16 2>&1 | grep -E "Msamples|instrumented" || exit 1; }
run PRGPU_NOOP=1
run PRGPU_PP_OCCUPANCY=2 PRGPU_PP_BLOCKS_PER_CU=2
run PRGPU_PP_OCCUPANCY=2 PRGPU_PP_BLOCKS_PER_CU=2 PRGPU_PP_SLOTS=768
run PRGPU_PP_OCCUPANCY=2 PRGPU_PP_BLOCKS_PER_CU=2 PRGPU_PP_SLOTS=1024
run PRGPU_PP_OCCUPANCY=2 PRGPU_PP_BLOCKS_PER_CU=2 PRGPU_PP_SLOTS=1024 PRGPU_PP_REFILL=56
