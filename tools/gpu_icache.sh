#!/bin/bash
# Instruction-cache counters of the persistent kernel (is the shading body's code footprint a cost?): usage  bash tools/gpu_icache.sh TAG "c4 c5" [LIBSUFFIX|default ...]
TAG=${1:-icache}; WL=${2:-c4}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | sort -u > "$OUT/${TAG}_avail.txt"
for L in "${@:-default}"; do
  if [ "$L" = "default" ]; then unset PRGPU_LIBRARY; else export PRGPU_LIBRARY=$R/pearray_amd/csrc/libprgpu_$L.so; fi
  for W in $WL; do
    if [ "$W" = "c4" ]; then PROG="python3 $R/bench.py --steps 8 --warmup 1 --profile-only"; else PROG="python3 $R/tools/profile_scene.py $W 8"; fi
    for SET in "ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "if SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
      set -- $SET; NAME=$1; shift
      D="$OUT/${TAG}_${L}_${W}_${NAME}"
      timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D" -- $PROG > "$D.log" 2>&1 || echo "pass $L $W $NAME failed"
      python3 - "$D" "$L" "$W" <<'PY' | tee -a "$OUT/${TAG}.txt"
import csv, glob, sys, collections
d, lib, w = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_path_persistent" in k:
            acc[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    print(lib, w, k)
    for c, x in sorted(v.items()):
        print("   %-28s %.6g" % (c, x))
PY
      find "$D" -name "*.csv" -size +5M -delete
    done
  done
done
