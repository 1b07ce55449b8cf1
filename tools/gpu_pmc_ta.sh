#!/bin/bash
# Texture-path (TA / TCP) counters of the bench workload, one rocprofv3 --pmc pass per small counter set (separate from any
# trace/stats pass).  usage (through gpurun, from the repo root):  bash tools/gpu_pmc_ta.sh TAG
TAG=${1:-ta}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
# (a pass with the TA_*_STALLED_BY_* counters aborted inside rocprofv3 on this pool and is left out)
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/${TAG}_ta$i" -- python3 "$R/bench.py" --steps 7 --warmup 1 --profile-only > "$OUT/${TAG}_ta$i.log" 2>&1 \
    || echo "pass $i ($set) failed or timed out" | tee -a "$OUT/${TAG}_ta_progress.log"
  echo "pass $i done: $set" | tee -a "$OUT/${TAG}_ta_progress.log"
done
find "$OUT" -path "*${TAG}_ta*" -name "*.csv" -size +20M -delete
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, json, sys
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); n = set()
for f in glob.glob("%s/%s_ta*/*/*_counter_collection.csv" % (out, tag)):
    for r in csv.DictReader(open(f)):
        if "k_path_persistent" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add((f, r["Dispatch_Id"]))
json.dump({"kernel": "k_path_persistent*", "iterations_per_pass": 8, "counters_sum_over_pass": dict(sorted(agg.items()))}, open("%s/%s_ta_summary.json" % (out, tag), "w"), indent=1)
for k, v in sorted(agg.items()):
    print("%-40s %.6g" % (k, v))
PY
