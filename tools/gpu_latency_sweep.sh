# The latency organisation of the persistent kernel (PRGPU_PP_KERNEL=latency) against the throughput one on one rank's 1/8 share, with its
# shading / refill thresholds swept.  usage (GPU box): bash tools/gpu_latency_sweep.sh c4|c5 out.log
export PRGPU_LIBRARY=${PRGPU_LIBRARY:-$PWD/pearray_amd/csrc/libprgpu.so}
WL=${1:-c4}; L=${2:-gpurun_out/r05/lat_sweep_$WL.log}
mkdir -p $(dirname $L); : > $L
echo "== throughput kernel" >> $L
PRGPU_PP_KERNEL=throughput python tools/gpu_counters.py 8 48 $WL >> $L 2>&1
for sm in 16 32 48 64; do for rf in 24 40 56; do
  echo "== latency PRGPU_PL_SHADE_MIN=$sm PRGPU_PL_REFILL=$rf" >> $L
  PRGPU_PP_KERNEL=latency PRGPU_PL_SHADE_MIN=$sm PRGPU_PL_REFILL=$rf python tools/gpu_counters.py 8 48 $WL >> $L 2>&1
done; done
grep "==\|ms/iteration" $L | cut -c1-200
