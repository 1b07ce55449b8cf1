#!/bin/bash
# Line coverage of the CPU checker (oracle/pr_oracle.cpp) under the test suite: which of the restated reference paths the tests reach at all.
# The checker restates the device code function by function and every GPU parity test runs both on the same scene, so a checker line no test
# reaches is a device path no test reaches.  usage (on the GPU box, through gpurun):  bash tools/oracle_coverage.sh TAG ["pytest selection"]
# Writes gpurun_out/TAG_oracle_gcov.txt (per-function summary) and gpurun_out/TAG_oracle_uncovered.txt (the lines never executed).
TAG=${1:-cov}; SEL=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd "$R/oracle" || exit 1
rm -f pr_oracle_cov.* libpr_oracle.so
g++ -O1 --coverage -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -c pr_oracle.cpp -o pr_oracle_cov.o || exit 1
g++ --coverage -shared -pthread pr_oracle_cov.o -o libpr_oracle.so || exit 1
cd "$R"
eval "timeout -k 10 1500 python3 -m pytest tests -q $SEL" > "$OUT/${TAG}_oracle_tests.txt" 2>&1   # (straight into the file: a pipe would hold the output back)
tail -3 "$OUT/${TAG}_oracle_tests.txt"
cd "$R/oracle"
gcov -f -o . pr_oracle_cov.o > "$OUT/${TAG}_oracle_gcov.txt" 2>&1
grep -n "#####" pr_oracle.cpp.gcov | cut -c1-200 > "$OUT/${TAG}_oracle_uncovered.txt"
tail -4 "$OUT/${TAG}_oracle_gcov.txt"
wc -l "$OUT/${TAG}_oracle_uncovered.txt"
rm -f pr_oracle_cov.* *.gcov libpr_oracle.so
make libpr_oracle.so > /dev/null
