#!/bin/bash
# Round-1 measurement script (run through gpurun): bench line + rocprofv3 kernel trace + PMC passes.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python bench.py --steps 96 --warmup 8 > $OUT/bench_r01.json 2> $OUT/bench_r01.err
tail -3 $OUT/bench_r01.err
cat $OUT/bench_r01.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 $R/bench.py --steps 16 --warmup 2 --profile-only > $OUT/prof_trace.log 2>&1
tail -2 $OUT/prof_trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --profile-only > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 $R/bench.py --steps 4 --warmup 1 --profile-only > $OUT/prof_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/prof_l2 -- python3 $R/bench.py --steps 4 --warmup 1 --profile-only > $OUT/prof_l2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/prof_sq -- python3 $R/bench.py --steps 4 --warmup 1 --profile-only > $OUT/prof_sq.log 2>&1
find $OUT -name "*.csv" | head -30
du -sh $OUT
