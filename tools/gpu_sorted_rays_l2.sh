#!/bin/bash
# L2 hit rate of the lockstep closest-hit traversal kernel with and without the sorted ray lists (rocprofv3 --pmc, its own passes).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for S in 0 1; do
  export PRGPU_MODE=lockstep PRGPU_SORT_RAYS=$S
  D=$OUT/sortl2_$S
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $D -- python3 $R/tools/gpu_sorted_rays.py 4 render-only > $D.log 2>&1 || echo "pass $S failed"
  python3 - $D $S <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        for name in ("k_trace_closest", "k_trace_shadow", "k_shade"):
            if name in k:
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print("PRGPU_SORT_RAYS=%s %-16s TCC_HIT %.4g TCC_MISS %.4g hit rate %.3f" % (sys.argv[2], k, h, m, h / max(h + m, 1)))
PY
  find $D -name "*.csv" -size +5M -delete
done
