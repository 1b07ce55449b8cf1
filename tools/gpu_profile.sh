#!/bin/bash
# Profile the bench workload on the GPU box: one rocprofv3 kernel-trace pass (timing) and separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE, L2 hit/miss, SQ counters), each with --kernel-trace only.  Run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash tools/gpu_profile.sh r01_v8'
# then, back in the container:  python tools/profile_to_summary.py r01_v8 8   (8 = iterations rendered in each --pmc pass)
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 "$R/bench.py" --steps 16 --warmup 8 --profile-only > "$OUT/${TAG}_trace.log" 2>&1
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_fetch" -- python3 "$R/bench.py" --steps 7 --warmup 1 --profile-only > "$OUT/${TAG}_fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_write" -- python3 "$R/bench.py" --steps 7 --warmup 1 --profile-only > "$OUT/${TAG}_write.log" 2>&1
echo "write pass done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/${TAG}_l2" -- python3 "$R/bench.py" --steps 7 --warmup 1 --profile-only > "$OUT/${TAG}_l2.log" 2>&1
echo "l2 pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d "$OUT/${TAG}_sq" -- python3 "$R/bench.py" --steps 7 --warmup 1 --profile-only > "$OUT/${TAG}_sq.log" 2>&1
echo "sq pass done"
# keep only the small CSVs (the merge back is limited to 64 MiB)
find "$OUT" -path "*${TAG}_*" -name "*.csv" -size +20M -delete
