#!/bin/bash
# Issue / texture-path occupancy of the persistent kernel on a tile share of the C4 frame (rocprofv3 --pmc, separate passes, --kernel-trace only):
#   bash tools/gpu_pmc_share.sh <world> [iterations]      e.g. 8 48 = rank 0's share of an 8-rank job
WORLD=${1:-8}; ITERS=${2:-48}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for SET in "sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU" "ta TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  set -- $SET; NAME=$1; shift
  D="$OUT/share${WORLD}_${NAME}"
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D" -- python3 $R/tools/gpu_counters.py $WORLD $ITERS c4 > "$D.log" 2>&1 || echo "pass $NAME failed"
  python3 - "$D" $ITERS <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_path_persistent_occ3<false" in k:      # the plain kernel: warm-up (8 iterations) + the timed launch (the instrumented launch has its own name)
            acc[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"]); n[k[:60]].add(row["Dispatch_Id"])
it = 8 + int(sys.argv[2])
for k, v in acc.items():
    print(k, "launches", len(n[k]), "iterations", it)
    for c, x in sorted(v.items()):
        print("   %-26s %.5g per iteration" % (c, x / it))
    if "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8 / it
        print("   -> %.3g shader cycles per iteration; TA busy %.3f; L2 hit %.3f" % (cyc, v["TA_TA_BUSY_sum"] / it / 256 / cyc, v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
    if "SQ_INSTS_VALU" in v:
        print("   -> VALU instructions per iteration %.4g (x 4 cycles / 1024 SIMDs = %.4g cycles); issuing %.3f / waiting %.3f of wave cycles" % (
            v["SQ_INSTS_VALU"] / it, v["SQ_INSTS_VALU"] / it * 4 / 1024, v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
  find "$D" -name "*.csv" -size +5M -delete
done
