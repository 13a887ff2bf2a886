#!/bin/bash
# Round-4 measurement set on ONE box: the driver's bench command, the same command under rocprofv3 --kernel-trace --stats,
# the two HBM-traffic PMC passes over the RLGC probe (bench-like emitter image), the idle gaps between the iteration's
# kernels.  Everything lands under gpurun_out/ (copy what is to be kept into profiles/).  Steps are chained with && so
# that a failed GPU step ends the call.
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 560 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4_final_bench.json 2> $O/r4_final_bench.err && echo "bench ok" &&
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof &&
M3D_BENCH_WALL_S=300 timeout -k 10 560 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 1 --no-cpu-baseline > $O/r4_final_bench_rocprof.json 2> $O/r4_final_bench_rocprof.err && echo "rocprof ok" &&
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/r4_final_kernel_stats.csv \; &&
export M3D_PROBE_IMAGE=emitters &&
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc; timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -o p -- python3 $R/scripts/gpu_probe_rlgc.py 64 2048 2048 2 > $O/r4_final_pmc_$c.log 2>&1 || exit 1
  find /tmp/pmc -name "*counter_collection.csv" -exec cp {} $O/r4_final_pmc_$c.csv \;
done && echo "pmc ok" && bash $R/scripts/gpu_iteration_gaps.sh > $O/r4_final_iteration_gaps.txt 2>&1 && echo "gaps ok" && ls -la $O | grep r4_final_
