#!/bin/bash
# VGPRs / scratch / LDS of every kernel of one translation unit:  bash scripts/resource_usage.sh kernels_tasks [filter]
# (the flags of bayeslogit_amd/build.py; machine-LICM off for the files built that way)
cd "$(dirname "$0")/../bayeslogit_amd/csrc" || exit 1
f=$1
extra=""
case $f in kernels_tasks|kernels_pg|kernels_beta|kernels_sweep1|kernels_sweep256) extra="-mllvm -disable-machine-licm";; esac
case $f in kernels_beta) extra="$extra -mllvm -amdgpu-promote-alloca-to-vector-vgpr-ratio=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/ru_$f.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass.*//' |
  awk '/Function Name/{if(n)print n, v; n=$3; v=""; next} {v=v" | "$0} END{print n, v}' | grep -E "${2:-.}" | while read -r name rest; do echo "$(echo $name | c++filt | cut -c1-70) $rest"; done
