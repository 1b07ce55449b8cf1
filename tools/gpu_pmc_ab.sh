#!/bin/bash
# A/B of two builds of libprgpu under rocprofv3 --pmc (SQ and TA counter sets only): usage  bash tools/gpu_pmc_ab.sh TAG [LIBSUFFIX ...]
# ("" = the shipped libprgpu.so, "ns" = pearray_amd/csrc/libprgpu_ns.so, ...).  Summaries: gpurun_out/TAG_<suffix>_{sq,ta}.txt
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" = "default" ]; then unset PRGPU_LIBRARY; else export PRGPU_LIBRARY=$R/pearray_amd/csrc/libprgpu_$L.so; fi
  for SET in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU" "ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    set -- $SET; NAME=$1; shift
    D="$OUT/${TAG}_${L}_${NAME}"
    timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D" -- python3 $R/bench.py --steps 7 --warmup 1 --profile-only > "$D.log" 2>&1 || echo "pass $L $NAME failed"
    python3 - "$D" "$L" "$NAME" <<'PY' > "$OUT/${TAG}_${L}_${NAME}.txt"
import csv, glob, sys, collections
d, lib, name = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_path_persistent" in k:
            acc[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    print(lib, name, k)
    for c, x in sorted(v.items()):
        print("   %-28s %.6g" % (c, x))
PY
    cat "$OUT/${TAG}_${L}_${NAME}.txt"
    find "$D" -name "*.csv" -size +5M -delete
  done
done
