#!/bin/bash
# Active-lane fraction of k_rpg_devroye by sampler class: scripts/gpu_c2.py draws z ~ U(0,4) (23 launches), then z in (0,3)
# (all |z|/2 < 1/t: 5 launches), then z in (3.2,4) (all >= 1/t: 5 launches); one --pmc pass, dispatches in that order.
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_c2lanes
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_c2lanes -- python3 scripts/gpu_c2.py > gpurun_out/c2lanes.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_c2lanes/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_rpg_devroye" not in r["Kernel_Name"]:
        continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(rows)
groups = {"z~U(0,4)": ids[3:23], "z in (0,3): class 1": ids[23:28], "z in (3.2,4): class 2": ids[28:33]}
for name, g in groups.items():
    m = {c: sum(rows[i][c] for i in g) / len(g) for c in rows[g[0]]}
    print(f"{name:26s} launches {len(g):2d}  wave-insts {m['SQ_INSTS_VALU']:.4g}  active-lane frac "
          f"{m['SQ_THREAD_CYCLES_VALU'] / (64 * m['SQ_ACTIVE_INST_VALU']):.3f}  VALU-active cycles/wave-inst "
          f"{m['SQ_ACTIVE_INST_VALU'] / m['SQ_INSTS_VALU']:.2f}")
PY
grep "ms" gpurun_out/c2lanes.log
