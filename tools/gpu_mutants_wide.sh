#!/bin/bash
# mutants of the six-wide traversal step (profiles/r05_mutate_wide.patch) against the tests that force six-wide trees
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05/mutations_wide.log
mkdir -p $R/gpurun_out/r05
: > $OUT
for m in 1 2 3; do
  case $m in
    1) what="the sixth child of a record never passes the box test";;
    2) what="a six-wide step makes room for three pushes instead of five (entries at the window's end overwritten)";;
    3) what="near and far z planes of children 4, 5 not swapped for rays with a negative z direction";;
  esac
  echo "=== mutant $m: $what" >> $OUT
  PRGPU_LIBRARY=$R/pearray_amd/csrc/libprgpu_mut$m.so timeout -k 10 900 python -m pytest tests/test_gpu_bvh_width.py -q 2>&1 | grep -E "^FAILED|passed|failed" | sed 's/ - .*//' >> $OUT
done
cat $OUT
