#!/bin/bash
# Profile a workload on the GPU box: one rocprofv3 kernel-trace pass (timing) and separate --pmc passes (FETCH_SIZE, WRITE_SIZE,
# L2 hit/miss, SQ counters), each with --kernel-trace only.  Run through gpurun from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/gpu_profile.sh r02 c4'      (c4 = bench.py; c5 | rough | metal_all = tools/profile_scene.py)
# then, back in the container:  python tools/profile_to_summary.py r02 8   (8 = iterations the PLAIN kernel renders in each --pmc pass)
TAG=${1:-prof}
WHAT=${2:-c4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = "c4" ]; then
  TRACE="python3 $R/bench.py --steps 16 --warmup 8 --profile-only"
  PMC="python3 $R/bench.py --steps 8 --warmup 1 --profile-only"   # (the warm-up launch is the scene's calibration launch: the instrumented variant; the plain kernel renders 8 iterations)
else
  TRACE="python3 $R/tools/profile_scene.py $WHAT 16"
  PMC="python3 $R/tools/profile_scene.py $WHAT 8"
fi
run() { # name, rocprofv3 arguments..., program
  local name=$1; shift
  timeout -k 10 240 rocprofv3 "$@" > "$OUT/${TAG}_${name}.log" 2>&1 || echo "pass $name failed or timed out"
  echo "pass $name done"
}
run trace --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- $TRACE
run fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_fetch" -- $PMC
run write --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_write" -- $PMC
run l2 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/${TAG}_l2" -- $PMC
run sq --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d "$OUT/${TAG}_sq" -- $PMC
run ta --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/${TAG}_ta" -- $PMC
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d "$OUT/${TAG}_tcp" -- $PMC
# keep only the small CSVs (the merge back is limited to 64 MiB)
find "$OUT" -path "*${TAG}_*" -name "*.csv" -size +20M -delete
