#!/bin/bash
# Round-end measurement set: GPU tests, smoke, bench, rocprofv3 kernel stats of the bench, two PMC passes.
# Everything lands under gpurun_out/ (copy what is to be kept into profiles/).
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/gpu_all.log 2>&1; echo "pytest rc=$?"; grep -v "Provide a valid" $O/gpu_all.log | tail -2
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python3 $R/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench_rocprof.err; echo "rocprof rc=$?"
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc; timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -o p -- python3 $R/scripts/gpu_probe_rlgc.py 64 2048 2048 2 > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  find /tmp/pmc -name "*counter_collection.csv" -exec cp {} $O/pmc_$c.csv \;
done
ls -la $O | tail -12
