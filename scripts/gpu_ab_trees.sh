#!/bin/bash
# rocprofv3 kernel averages of the RLGC probe in several ab/<name> trees (scripts/make_ab_tree.sh) on ONE box.
# usage: gpu_ab_trees.sh "kernel regex" tree1 tree2 ...   (env passes through, e.g. M3D_PROBE_IMAGE=emitters); "." = the working tree
R=$(cd "$(dirname "$0")/.." && pwd); pat="$1"; shift
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  root=$R/ab/$t; [ "$t" = "." ] && root=$R
  rm -rf /tmp/abt_prof
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt_prof -o p -- python3 $root/scripts/gpu_probe_rlgc.py ${GEOM:-64 2048 2048} ${ITERS:-4} > /tmp/abt_out.txt 2>&1 || { echo "== $t FAILED"; tail -5 /tmp/abt_out.txt; continue; }
  python3 - "$t" "$pat" <<'PY'
import csv, sys, re, glob
tag, pat = sys.argv[1], sys.argv[2]
f = glob.glob('/tmp/abt_prof/**/p_kernel_stats.csv', recursive=True)[0]
tot = 0.0; hits = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Name"])
    if not m: continue
    ms = float(r["TotalDurationNs"]) / 1e6
    if re.search(r"k_(x_|y_|z_|kl_|fused_fin)", m.group(1)): tot += ms
    if re.search(pat, m.group(1)): hits.append(f"{m.group(1)[:44]} x{r['Calls']} avg {float(r['AverageNs'])/1e3:8.1f} us")
print(f"== {tag:8s} iteration kernels total {tot:8.2f} ms | " + " | ".join(hits))
PY
done
