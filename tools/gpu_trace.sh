#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-t}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --steps 8 --warmup 2 --profile-only > $OUT/${TAG}_trace.log 2>&1
grep -h "Msamples" $OUT/${TAG}_trace.log | cut -c1-160
