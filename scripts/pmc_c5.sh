#!/bin/bash
# PMC passes over scripts/gpu_c5.py: where the P = 256 X'Omega X kernel's cycles go.
set -o pipefail
export TMPDIR=/tmp BL_N=${BL_N:-2000000}
mkdir -p gpurun_out
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc5_$tag
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc5_$tag -- python3 scripts/gpu_c5.py > gpurun_out/pmc5_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/pmc5_$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections, re
for f in sorted(glob.glob('gpurun_out/pmc5_*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        if m and 'xwx' in m.group(1):
            acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(k, {c: '%.4g' % (sum(v) / len(v)) for c, v in cs.items()}, 'n', len(next(iter(cs.values()))))
PY
