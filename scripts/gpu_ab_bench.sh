#!/bin/bash
# Same-box A/B of the iteration's kernels under the BENCH workload: rocprofv3 kernel averages of a short bench.py run in
# each ab/<name> tree (scripts/make_ab_tree.sh, plus bench.py and __graft_entry__.py copied into the tree).
# usage: gpu_ab_bench.sh "kernel regex" tree1 tree2 ...
R=$(cd "$(dirname "$0")/.." && pwd); pat="$1"; shift
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  root=$R/ab/$t
  rm -rf /tmp/abb_prof
  M3D_BENCH_WALL_S=300 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abb_prof -o p -- python3 $root/bench.py --gpus 1 --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline > /tmp/abb_out.txt 2>/tmp/abb_err.txt || { echo "== $t FAILED"; tail -5 /tmp/abb_err.txt; exit 1; }
  python3 - "$t" "$pat" <<'PY'
import csv, sys, re, glob, json
tag, pat = sys.argv[1], sys.argv[2]
d = json.loads([l for l in open('/tmp/abb_out.txt') if l.startswith('{')][-1])
f = glob.glob('/tmp/abb_prof/**/p_kernel_stats.csv', recursive=True)[0]
hits = []; tot = 0.0; n_it = 0
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Name"])
    if m and re.search(r"k_(x_|y_|z_|kl_|fused_fin)", m.group(1)): tot += float(r["TotalDurationNs"]) / 1e6
    if m and "split_chain" in m.group(1): n_it = int(r["Calls"])
    if m and re.search(pat, m.group(1)): hits.append(f"{m.group(1)[:40]} x{r['Calls']} avg {float(r['AverageNs'])/1e3:8.1f} us")
wp = d.get("workspace_placement", {}).get("candidates_ms_rank0")
if wp: print(f"   {tag:8s} workspace candidates {wp}")
print(f"== {tag:8s} value {d['value']} frac {d['roofline']['frac']} iteration kernels {tot / max(n_it, 1):.3f} ms | " + " | ".join(hits))
PY
done
