#!/bin/bash
# A/B of the RLGC iteration's kernels between the in-tree library and scratch_libs/$1 on ONE box (usage: gpu_ab_iter.sh lib.so [tagA tagB])
R=$(cd "$(dirname "$0")/.." && pwd); L=$R/merfish3d-analysis_amd/csrc/libm3d_hip.so
cp $L /tmp/ab_iter_orig.so
bash $R/scripts/gpu_kstats.sh ${2:-intree} 64 2048 2048 6 | grep -v "^kstats" > /dev/null
cp $R/scratch_libs/$1 $L
bash $R/scripts/gpu_kstats.sh ${3:-other} 64 2048 2048 6 | grep -v "^kstats" > /dev/null
cp /tmp/ab_iter_orig.so $L
python3 - "$R/gpurun_out/kstats_${2:-intree}.txt" "$R/gpurun_out/kstats_${3:-other}.txt" <<'PY'
import sys,re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r"(\S.*?)\s+calls=\s*(\d+)\s+total_ms=\s*([\d.]+)\s+avg_us=\s*([\d.]+)", l)
        if m: d[m.group(1)]=(int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return d
a,b=load(sys.argv[1]),load(sys.argv[2])
ta=tb=0
for k in sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0,0))[1])):
    ca,ma,ua=a.get(k,(0,0,0)); cb,mb,ub=b.get(k,(0,0,0))
    print(f"{k:44s} A: {ca:4d} x {ua:8.1f} us = {ma:8.2f} ms   B: {cb:4d} x {ub:8.1f} us = {mb:8.2f} ms")
    ta+=ma; tb+=mb
print(f"total k_* ms: A {ta:.2f}  B {tb:.2f}")
PY
