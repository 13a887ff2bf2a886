#!/bin/bash
# EA (fabric-side) read requests of the RLGC probe's kernels with and without the XCD pairing of the X passes' row pairs
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
export M3D_PROBE_IMAGE=emitters
for xp in 1 0; do
  rm -rf /tmp/pmc_xp
  M3D_FFT_XCDPAIR=$xp timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/pmc_xp -o p -- python3 $R/scripts/gpu_probe_rlgc.py 64 2048 2048 2 > $O/pmc_xp_$xp.log 2>&1
  echo "xcdpair $xp rc=$?"
  find /tmp/pmc_xp -name "*counter_collection.csv" -exec cp {} $O/pmc_xp_$xp.csv \;
done
