#!/bin/bash
# L2 (TCC) and L1 (TCP) counter passes over the RLGC probe, one rocprofv3 --pmc run per pass (counters only: no trace
# domains), program directly after "--".  Raw CSVs land in gpurun_out/pmc_mem_<pass>.csv; scripts/pmc_sq_summary.py
# reduces them per kernel (mean per dispatch).  usage: gpu_pmc_mem.sh [Z Y X iters]
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
Z=${1:-64}; Y=${2:-2048}; X=${3:-2048}; IT=${4:-2}
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
  local name=$1; shift
  rm -rf /tmp/pmc_mem
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_mem -o p -- python3 $R/scripts/gpu_probe_rlgc.py $Z $Y $X $IT > $O/pmc_mem_$name.log 2>&1
  echo "pmc pass $name rc=$?"
  find /tmp/pmc_mem -name "*counter_collection.csv" -exec cp {} $O/pmc_mem_$name.csv \;
}
pass l2a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum &&
pass l2b TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_READ_sum &&
pass l2c TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_EA0_WRREQ_STALL_sum &&
pass l1a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum &&
pass l1b TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCC_TAG_STALL_sum
ls -la $O/pmc_mem_*.csv
