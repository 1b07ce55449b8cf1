#!/bin/bash
# usage: gpu_pmc.sh TAG "COUNTER COUNTER ..."  -> gpurun_out/TAG_pmc/
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $@ --kernel-trace --output-format csv -d $OUT/${TAG}_pmc -- python3 $R/bench.py --steps 3 --warmup 1 --profile-only > $OUT/${TAG}_pmc.log 2>&1
tail -1 $OUT/${TAG}_pmc.log | cut -c1-120
