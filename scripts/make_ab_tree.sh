#!/bin/bash
# Snapshot the working tree's package (sources + the library built from them) as ab/<name>/ for A/B timing on ONE GPU
# box: every tree carries its own hash-checked libm3d_hip.so, nothing is ever copied over the in-tree library.
# usage: scripts/make_ab_tree.sh <name>   then on the box e.g.  python3 ab/<name>/scripts/gpu_probe_rlgc.py 64 2048 2048 4
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/ab/$1
[ -n "$1" ] || { echo "usage: $0 name"; exit 1; }
python3 $R/merfish3d-analysis_amd/csrc/build.py > /dev/null
rm -rf $D; mkdir -p $D/scripts
cp -r $R/include $D/ && mkdir -p $D/merfish3d-analysis_amd && (cd $R/merfish3d-analysis_amd && tar cf - --exclude=.objcache --exclude=.build --exclude=__pycache__ .) | (cd $D/merfish3d-analysis_amd && tar xf -)
cp $R/scripts/gpu_probe_rlgc.py $R/scripts/gpu_probe_split.py $R/scripts/gpu_probe_decode.py $D/scripts/ 2>/dev/null || true
cp $R/merfish3d_analysis_amd.py $D/ 2>/dev/null || true
echo "ab tree $D: $(python3 -c "import sys; sys.path.insert(0,'$D'); import importlib; print(importlib.import_module('merfish3d-analysis_amd._buildcheck').embedded_source_hash('$D/merfish3d-analysis_amd/csrc/libm3d_hip.so')[:16])")"
