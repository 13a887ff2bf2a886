#!/bin/bash
# scratch: per-dispatch kernel durations of a short RLGC probe (which of the four convolutions costs what)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python3 $R/scripts/gpu_probe_rlgc.py 64 2048 2048 3 > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
    if m:
        seq.append((m.group(1), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
# last full iteration: from the last k_split_ratio backwards/forwards
idx = [i for i, (n, _) in enumerate(seq) if n.startswith("k_split_ratio")]
lo = idx[-3] if len(idx) >= 3 else 0
hi = idx[-2] if len(idx) >= 2 else len(seq)
# print one iteration's dispatches in order, starting at the Hu convolution's first kernel (just after the previous update)
for n, d in seq[lo:hi]:
    print(f"{d:8.3f} ms  {n}")
PY
