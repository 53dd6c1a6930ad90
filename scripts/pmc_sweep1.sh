#!/bin/bash
# PMC passes over scripts/gpu_sweep1.py (one case): where the single-pass sweep kernel's cycles go.
set -o pipefail
export TMPDIR=/tmp BL_CASES=1
mkdir -p gpurun_out
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc1_$tag
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc1_$tag -- python3 scripts/gpu_sweep1.py > gpurun_out/pmc1_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/pmc1_$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc1_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            if 'sweep' in k or 'psi_omega' in k or 'xwx' in k:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in acc.items():
            print(k, {c: '%.4g' % (sum(v) / len(v)) for c, v in cs.items()}, 'launches', len(next(iter(cs.values()))))
PY
