# knob sweep of small tile shares and the full frame (tools/gpu_probe_share8.py); usage: bash tools/gpu_sweep_share8.sh
run() { w=$1; shift; echo "== $*"; env "$@" timeout -k 10 100 python tools/gpu_probe_share8.py $w 32 || exit 1; }
for w in 1 2 4 8 16 32; do run $w A=1; done
run 4 PRGPU_PP_SLOTS=704
run 4 PRGPU_PP_SLOTS=704 PRGPU_PP_SHADER=0
run 4 PRGPU_PP_SLOTS=384
run 2 PRGPU_PP_SLOTS=384
run 1 PRGPU_PP_SLOTS=384
run 1 PRGPU_PP_SLOTS=448
