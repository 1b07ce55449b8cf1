#!/bin/bash
# quick profile: kernel trace + SQ/L2 PMC passes on a short bench run; outputs under gpurun_out/q_*
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-q}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --steps 8 --warmup 2 --profile-only > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/${TAG}_sq -- python3 $R/bench.py --steps 3 --warmup 1 --profile-only > $OUT/${TAG}_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/${TAG}_l2 -- python3 $R/bench.py --steps 3 --warmup 1 --profile-only > $OUT/${TAG}_l2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --profile-only > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 $R/bench.py --steps 3 --warmup 1 --profile-only > $OUT/${TAG}_write.log 2>&1
grep -h "Msamples" $OUT/${TAG}_trace.log | cut -c1-200
