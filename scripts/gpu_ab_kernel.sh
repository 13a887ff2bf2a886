#!/bin/bash
# scratch: A/B one kernel's rocprofv3 average between the in-tree library and scratch_libs/$1 (usage: gpu_ab_kernel.sh lib.so pattern)
cd "$(dirname "$0")/.."
L=$PWD/merfish3d-analysis_amd/csrc/libm3d_hip.so
cp $L /tmp/orig.so
run() {
  rm -rf /tmp/ab_prof; (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -o p -- python3 $OLDPWD/scripts/gpu_probe_rlgc.py 64 2048 2048 4 > /dev/null 2>&1)
  grep -E "$2" /tmp/ab_prof/p_kernel_stats.csv | awk -F'",' '{print $1}' | cut -c1-70 | paste -d' ' - <(grep -E "$2" /tmp/ab_prof/p_kernel_stats.csv | awk -F, '{print $(NF-6), $(NF-4)}')
}
echo "== in-tree"; run x "$2"
cp scratch_libs/$1 $L
echo "== $1"; run x "$2"
cp /tmp/orig.so $L
