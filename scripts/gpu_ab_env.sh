#!/bin/bash
# scratch: rocprofv3 average of one kernel under different env settings (usage: gpu_ab_env.sh pattern "ENV=.." ...)
cd "$(dirname "$0")/.."
pat="$1"; shift
for cfg in "$@"; do
  rm -rf /tmp/ab_prof
  (cd /tmp && env $cfg TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -o p -- python3 $OLDPWD/scripts/gpu_probe_rlgc.py 64 2048 2048 4 > /tmp/ab_out.txt 2>&1)
  echo "== $cfg: $(grep -E "$pat" /tmp/ab_prof/p_kernel_stats.csv | awk -F, '{printf "%s calls avg %.3f ms; ", $(NF-6), $(NF-4)/1e6}') $(grep 'iters=8' /tmp/ab_out.txt | sed 's/.*wall=/wall=/' | cut -c1-12)"
done
