#!/bin/bash
# rocprofv3 kernel trace + SQ counters of rank 0's 1/8 tile share of the C4 frame (32 iterations in one launch), next to the share probes:
#   gpurun --timeout 900 -- 'bash tools/gpu_profile_share8.sh r02_share8'
TAG=${1:-share8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 $R/tools/profile_scene.py share8 32 > "$OUT/${TAG}_trace.log" 2>&1 || echo "trace pass failed"
echo "trace pass done"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d "$OUT/${TAG}_sq" -- python3 $R/tools/profile_scene.py share8 32 > "$OUT/${TAG}_sq.log" 2>&1 || echo "sq pass failed"
echo "sq pass done"
cd "$R"
for w in 1 2 4 8 16 32; do timeout -k 10 100 python tools/gpu_probe_share8.py $w 32 || exit 1; done > "$OUT/${TAG}_shares.log" 2>&1
PRGPU_PP_SHADER=0 timeout -k 10 100 python tools/gpu_probe_share8.py 8 32 >> "$OUT/${TAG}_shares.log" 2>&1
timeout -k 10 100 python tools/gpu_block_life.py 8 32 > "$OUT/${TAG}_block_life.log" 2>&1
cat "$OUT/${TAG}_shares.log" "$OUT/${TAG}_block_life.log"
find "$OUT" -path "*${TAG}_*" -name "*.csv" -size +20M -delete
