# knob sweep (tools/gpu_probe_share8.py): usage: bash tools/gpu_sweep_share8.sh [world] [iterations]
W=${1:-1}; I=${2:-64}
run() { echo -n "$*: "; env "$@" timeout -k 10 100 python tools/gpu_probe_share8.py $W $I | cut -c1-48 || exit 1; }
run A=1
run PRGPU_PP_REFILL=40
run PRGPU_PP_REFILL=44
run PRGPU_PP_REFILL=52
run PRGPU_PP_REFILL=56
run PRGPU_PP_SHADE_PARTIAL=8
run PRGPU_PP_SHADE_PARTIAL=24
run PRGPU_PP_SHADE_PARTIAL=32
run PRGPU_PP_PARTIAL_ACT=48
run PRGPU_PP_PARTIAL_ACT=56
run PRGPU_PP_LEAF_BIAS=90
run PRGPU_PP_LEAF_BIAS=110
run PRGPU_PP_LEAF_BIAS=125
run PRGPU_PP_REFILL_MIN=4
run PRGPU_PP_BOTH=8
run A=2
