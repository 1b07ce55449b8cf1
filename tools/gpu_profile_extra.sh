#!/bin/bash
# Extras of a profile round: (a) the sorted-ray-push knob (PRGPU_PP_SORT=1) on the C4 bench: kernel trace + L2 hit/miss + SQ passes,
# (b) the raw per-lane record gather rate of the chip (tools/micro/gather_bench.hip).  usage: bash tools/gpu_profile_extra.sh TAG
TAG=${1:-extra}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PRGPU_PP_SORT=1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_sort_trace" -- python3 $R/bench.py --steps 16 --warmup 8 --profile-only > "$OUT/${TAG}_sort_trace.log" 2>&1; echo "sorted trace done"
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/${TAG}_sort_l2" -- python3 $R/bench.py --steps 7 --warmup 1 --profile-only > "$OUT/${TAG}_sort_l2.log" 2>&1; echo "sorted l2 done"
unset PRGPU_PP_SORT
PRGPU_PP_SORT=1 timeout -k 10 300 python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline > "$OUT/${TAG}_sort_bench.json" 2>/dev/null; echo "sorted bench done"
timeout -k 10 300 python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline > "$OUT/${TAG}_plain_bench.json" 2>/dev/null; echo "plain bench done"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/micro/gather_bench.hip -o /tmp/gather_bench && timeout -k 10 120 /tmp/gather_bench > "$OUT/${TAG}_gather_bench.txt" 2>&1; echo "gather bench done"; cat "$OUT/${TAG}_gather_bench.txt"
find "$OUT" -path "*${TAG}_*" -name "*.csv" -size +20M -delete
