#!/bin/bash
# Scratch (not a test): clocks, temperatures and power of the card sampled every 2 s while the RLGC probe iterates
# or the bench (does the slow state of the strided passes show in the memory clock or the HBM temperature?).  usage: gpu_smi_during_rlgc.sh [iters] | bench [tiles]
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/smi_during_rlgc.txt
mkdir -p $R/gpurun_out
echo "# before" > $O
rocm-smi --showclocks --showtemp --showpower --showperflevel >> $O 2>&1
( while true; do echo "## $(date +%s.%N)"; rocm-smi --showclocks --showtemp --showpower 2>&1 | grep -E "mclk|sclk|fclk|Temperature|Power|power" ; sleep 2; done ) >> $O &
SMI=$!
if [ "$1" = bench ]; then   # the driver's workload, $2 timed tiles
  M3D_BENCH_WALL_S=400 timeout -k 10 500 python3 $R/bench.py --gpus 1 --steps ${2:-3} --warmup 1 --no-cpu-baseline > $R/gpurun_out/smi_probe.txt 2>$R/gpurun_out/smi_probe.err
else
  M3D_PROBE_IMAGE=emitters timeout -k 10 400 python3 $R/scripts/gpu_probe_rlgc.py 64 2048 2048 ${1:-600} > $R/gpurun_out/smi_probe.txt 2>&1
fi
rc=$?
kill $SMI
wait $SMI 2>/dev/null
tail -3 $R/gpurun_out/smi_probe.txt | cut -c1-400
exit $rc
