#!/bin/bash
# the beta stage of the C5-shaped chain (BL_P, default 256; data-rich posterior) under the two kernels of the constrained sweeps: 1 = row-split segments of 64, 0 = one wavefront
mkdir -p gpurun_out
for m in ${BL_MODES:-1 0}; do
  echo "== BL_BETA_SPLIT=$m"
  BL_BETA_SPLIT=$m BL_N=${BL_N:-4000000} timeout -k 10 300 python scripts/gpu_c5.py 2>&1 | grep -v "^$" | tail -${BL_TAIL:-6} || exit 1
done
