#!/bin/bash
# Sweep of the persistent kernel's scheduling knobs inside ONE gpurun call (results never change, tests/test_gpu_parity.py):
#   bash tools/gpu_knob_sweep.sh TAG WORKLOAD "ENV1=a ENV2=b" "ENV1=c" ...      ("-" = the defaults)
# One line per setting in gpurun_out/TAG.log: Msamples/s, ms per step, lane utilisation, shade pass fill.
TAG=${1:-sweep}; W=${2:-c5}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
: > "$OUT/$TAG.log"
for S in "$@"; do
  if [ "$S" = "-" ]; then E=""; else E="$S"; fi
  LINE=$(env $E timeout -k 10 300 python3 $R/bench.py --workload $W --steps 24 --warmup 8 --no-cpu-baseline 2>>"$OUT/$TAG.err" | tail -1)
  echo "[$S] $(echo "$LINE" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); print(d["value"], d["ms_per_step"], r.get("lane_utilisation"), r.get("shade_pass_fill"))' 2>/dev/null)" | tee -a "$OUT/$TAG.log"
done
