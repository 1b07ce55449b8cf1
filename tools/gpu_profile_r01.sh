#!/bin/bash
# Round-1 measurement script (run through gpurun): bench line + rocprofv3 kernel trace + PMC passes.
# usage: gpu_profile_r01.sh TAG   (then: python tools/profile_to_summary.py TAG 8)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r01}
mkdir -p $OUT
cd $R
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cat $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --steps 16 --warmup 2 --profile-only > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 $R/bench.py --steps 7 --warmup 1 --profile-only > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 $R/bench.py --steps 7 --warmup 1 --profile-only > $OUT/${TAG}_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/${TAG}_l2 -- python3 $R/bench.py --steps 7 --warmup 1 --profile-only > $OUT/${TAG}_l2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/${TAG}_sq -- python3 $R/bench.py --steps 7 --warmup 1 --profile-only > $OUT/${TAG}_sq.log 2>&1
du -sh $OUT
