#!/bin/bash
# MFMA / VALU counters of the fused decode kernel (rocprofv3 --pmc, program directly after "--"); raw CSV ->
# gpurun_out/pmc_decode_<pass>.csv, reduced by scripts/pmc_sq_summary.py.  usage: gpu_pmc_decode.sh [Z Y X]
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
Z=${1:-16}; Y=${2:-2048}; X=${3:-2048}
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift; rm -rf /tmp/pmc_dec
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_dec -o p -- python3 $R/scripts/gpu_probe_decode.py $Z $Y $X 22 2 > $O/pmc_decode_$name.log 2>&1
  echo "pmc pass $name rc=$?"; find /tmp/pmc_dec -name "*counter_collection.csv" -exec cp {} $O/pmc_decode_$name.csv \; ; }
pass mfma SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE &&
pass wait SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
python3 $R/scripts/pmc_sq_summary.py $O/pmc_decode_mfma.csv $O/pmc_decode_wait.csv > $O/pmc_decode.json
tail -2 $O/pmc_decode_mfma.log | head -1
