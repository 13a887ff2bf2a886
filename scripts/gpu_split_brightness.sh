#!/bin/bash
# k_split_ratio's rocprofv3 average at several image brightnesses (usage: gpu_split_brightness.sh tag)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
tag=${1:-new}
cd /tmp && export TMPDIR=/tmp
: > $O/split_brightness_$tag.txt
for mean in ${COUNTS:-100 1000 10000 60000}; do
  rm -rf /tmp/sb_prof
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_prof -o p -- python3 $R/scripts/gpu_probe_split.py $mean 6 > /tmp/sb_out.txt 2>&1 || { echo "probe failed at $mean"; tail -5 /tmp/sb_out.txt; break; }
  f=$(find /tmp/sb_prof -name p_kernel_stats.csv | head -1)
  line=$(grep k_split_ratio $f | awk -F, '{printf "calls %s avg_ms %.3f min_ms %.3f max_ms %.3f", $(NF-6), $(NF-4)/1e6, $(NF-2)/1e6, $(NF-1)/1e6}')
  echo "$tag mean_count=$mean k_split_ratio: $line | $(grep '^mean' /tmp/sb_out.txt)" | tee -a $O/split_brightness_$tag.txt
done
