#!/bin/bash
# the constrained sweeps of 64 < P <= 256 under the two kernels (BL_BETA_SPLIT = 1 row-split segments of 64 / 0 one wavefront): same digest = same chain
mkdir -p gpurun_out
for m in 1 0; do
  echo "== BL_BETA_SPLIT=$m"
  BL_BETA_SPLIT=$m timeout -k 10 240 python scripts/gpu_beta.py 70 97 128 200 256 2>&1 | grep "constrain=1" || exit 1
done
