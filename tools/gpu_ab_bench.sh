#!/bin/bash
# A/B of builds of libprgpu inside ONE gpurun call (run-to-run spread between calls is +-4 %): bench.py per build and workload, alternating.
#   bash tools/gpu_ab_bench.sh TAG "c4 c5" default r3 [default r3 ...]     ("default" = the shipped libprgpu.so, X = libprgpu_X.so)
# One JSON line per run in gpurun_out/TAG.log, prefixed by the build and workload.
TAG=${1:-ab}; WL=${2:-c4}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
: > "$OUT/$TAG.log"
for E in "$@"; do
  L=${E%%@*}   # X@VAR=VALUE: build X with an environment variable set for its runs
  unset PRGPU_BVH_WIDTH PRGPU_PP_KERNEL
  if [ "$E" != "$L" ]; then export "${E#*@}"; fi
  if [ "$L" = "default" ]; then unset PRGPU_LIBRARY; else export PRGPU_LIBRARY=$R/pearray_amd/csrc/libprgpu_$L.so; fi
  for W in $WL; do
    LINE=$(timeout -k 10 300 python3 $R/bench.py --workload $W --steps 32 --warmup 8 --no-cpu-baseline 2>>"$OUT/$TAG.err" | tail -1)
    if [ -z "$LINE" ]; then   # the run died (a faulting kernel, a timeout): nothing else is started on this GPU in this call
      echo "$L $W FAILED: see $TAG.err; stopping" | tee -a "$OUT/$TAG.log"; exit 1
    fi
    echo "$E $W $LINE" >> "$OUT/$TAG.log"
    echo "$E $W $(echo "$LINE" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); print(d["value"], d["ms_per_step"], r.get("frac"), r.get("nodes_per_closest_ray"), r.get("leaves_per_closest_ray"), r.get("lane_utilisation"), r.get("bvh_width"), r.get("bvh_cost_estimate"))' 2>/dev/null)"
  done
done
