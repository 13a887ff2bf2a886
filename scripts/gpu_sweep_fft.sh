#!/bin/bash
# tuning sweep of the fused FFT convolution tile shapes (scratch tool)
run() { echo "== $*"; env "$@" timeout -k 10 120 python scripts/gpu_probe_rlgc.py 64 2048 2048 3 2>&1 | grep "iters=6"; }
run M3D_X=0
run M3D_FFT_YT=1024
run M3D_FFT_YT=768
run M3D_FFT_P=8 M3D_FFT_ZT=64
run M3D_FFT_P=16 M3D_FFT_ZT=64
