#!/bin/bash
# scratch: time alternative builds of libm3d_hip.so kept under scratch_libs/ (swapped in place on the GPU box copy)
cd "$(dirname "$0")/.."
L=merfish3d-analysis_amd/csrc/libm3d_hip.so
cp $L /tmp/orig.so
for f in "$@"; do
  echo "== $f"
  cp scratch_libs/$f $L
  timeout -k 10 120 python scripts/gpu_probe_rlgc.py 64 2048 2048 3 2>&1 | grep "iters=6" || echo "   (failed)"
done
cp /tmp/orig.so $L
echo "== default"; timeout -k 10 120 python scripts/gpu_probe_rlgc.py 64 2048 2048 3 2>&1 | grep "iters=6"
