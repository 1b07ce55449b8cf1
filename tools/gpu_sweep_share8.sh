# knob sweep of small tile shares and the full frame (tools/gpu_probe_share8.py); usage: bash tools/gpu_sweep_share8.sh
run() { w=$1; shift; echo "== $*"; env "$@" timeout -k 10 100 python tools/gpu_probe_share8.py $w 32 96 || exit 1; }
run 8 PRGPU_PP_LAYER_SPEED=1.15,1.35
run 8 PRGPU_PP_LAYER_SPEED=1.2,1.45
run 8 PRGPU_PP_LAYER_SPEED=1.3,1.6
run 8 PRGPU_PP_LAYER_SPEED=1.12,1.25
run 8 PRGPU_PP_TUNE_ORDER=0
PRGPU_PP_LAYER_SPEED=1.2,1.45 timeout -k 10 100 python tools/gpu_block_life.py 8 32
