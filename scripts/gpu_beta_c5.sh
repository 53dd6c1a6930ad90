#!/bin/bash
# the beta stage of the C5-shaped chain (P = 256, data-rich posterior) under the row-split kernels: 2 = segments of 64, 1 = blocks of 16
mkdir -p gpurun_out
for m in ${BL_MODES:-2 1}; do
  echo "== BL_BETA_SPLIT=$m"
  BL_BETA_SPLIT=$m BL_N=${BL_N:-4000000} timeout -k 10 300 python scripts/gpu_c5.py 2>&1 | grep -v "^$" | tail -${BL_TAIL:-6} || exit 1
done
