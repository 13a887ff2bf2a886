#!/bin/bash
# scratch: tile-shape sweep of the fused FFT engine (usage: gpu_sweep_fft.sh "ENV=.. ENV=.." ...)
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python scripts/gpu_probe_rlgc.py 64 2048 2048 3 2>&1 | grep "iters=6" || echo "   (failed)"
done
