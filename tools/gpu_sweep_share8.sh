# knob sweep of small tile shares and the full frame (tools/gpu_probe_share8.py); usage: bash tools/gpu_sweep_share8.sh
run() { w=$1; shift; echo "== $*"; env "$@" timeout -k 10 100 python tools/gpu_probe_share8.py $w 32 || exit 1; }
run 8 A=1
run 8 PRGPU_PP_BLOCKS_PER_CU=2 PRGPU_PP_SLOTS=512
run 8 PRGPU_PP_BLOCKS_PER_CU=2 PRGPU_PP_SLOTS=512 PRGPU_PP_OCCUPANCY=2
run 16 PRGPU_PP_BLOCKS_PER_CU=2
run 16 PRGPU_PP_BLOCKS_PER_CU=1
