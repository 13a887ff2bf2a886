#!/bin/bash
# SQ counter passes (occupancy / wave-wait / LDS conflicts / instruction mix) over the RLGC probe, one rocprofv3
# --pmc run per pass (8 SQ slots each), program directly after "--".  Raw CSVs land in gpurun_out/pmc_sq_<pass>.csv;
# reduce them with scripts/pmc_sq_summary.py.  usage: gpu_pmc_sq.sh [Z Y X iters]
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
Z=${1:-64}; Y=${2:-2048}; X=${3:-2048}; IT=${4:-2}
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
  local name=$1; shift
  rm -rf /tmp/pmc_sq
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_sq -o p -- python3 $R/scripts/gpu_probe_rlgc.py $Z $Y $X $IT > $O/pmc_sq_$name.log 2>&1
  echo "pmc pass $name rc=$?"
  find /tmp/pmc_sq -name "*counter_collection.csv" -exec cp {} $O/pmc_sq_$name.csv \;
}
pass wait SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE &&
pass lds SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
pass mem SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
ls -la $O/pmc_sq_*.csv
