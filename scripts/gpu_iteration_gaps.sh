#!/bin/bash
# Idle time between the kernels of the RLGC iteration: rocprofv3 kernel trace of the probe, gaps between consecutive
# dispatches by (previous kernel -> next kernel).  usage: gpu_iteration_gaps.sh [Z Y X iters]
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gaps_prof
M3D_PROBE_IMAGE=emitters timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_prof -o p -- python3 $R/scripts/gpu_probe_rlgc.py ${1:-64} ${2:-2048} ${3:-2048} ${4:-8} > $O/gaps_probe.log 2>&1
python3 - <<'PY'
import csv, glob, re, collections
f = glob.glob('/tmp/gaps_prof/**/p_kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
# the last solver call: take the final 16 iterations' worth of kernels
names = [n for _, _, n in rows]
last_split = [i for i, n in enumerate(names) if n == "k_x_inverse_split_chain"]
lo = last_split[-15] if len(last_split) >= 15 else 0
seg = rows[lo:]
gaps = collections.defaultdict(list)
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
    gaps[(n0, n1)].append(s1 - e0)
    busy += e0 - s0
span = seg[-1][1] - seg[0][0]
print(f"kernels in the window: {len(seg)}, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms ({100*(span-busy)/span:.2f} %)")
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"{a:32s} -> {b:32s} n={len(v):3d} mean gap {sum(v)/len(v)/1e3:8.1f} us   total {sum(v)/1e6:7.3f} ms")
PY
