#!/bin/bash
# Scratch: time the RLGC probe's kernels with each diagnostic build in scratch_libs/ swapped in (usage: gpu_variants.sh kernel_regex V1 V2 ...)
R=$(cd "$(dirname "$0")/.." && pwd)
pat=$1; shift
cp $R/merfish3d-analysis_amd/csrc/libm3d_hip.so /tmp/libm3d_base.so
for v in base "$@"; do
  [ $v = base ] && cp /tmp/libm3d_base.so $R/merfish3d-analysis_amd/csrc/libm3d_hip.so || cp $R/scratch_libs/libm3d_$v.so $R/merfish3d-analysis_amd/csrc/libm3d_hip.so
  bash $R/scripts/gpu_kstats.sh $v > /dev/null 2>&1
  echo "== $v"; grep -E "$pat" $R/gpurun_out/kstats_$v.txt
done
cp /tmp/libm3d_base.so $R/merfish3d-analysis_amd/csrc/libm3d_hip.so
