#!/bin/bash
# The P = 256 single-pass sweep on the full C5 shard (12.5e6 x 256): per-kernel times (--kernel-trace --stats) and HBM traffic
# (FETCH_SIZE, WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE counts half of a wide coalesced read: x 2), 12 sweeps of
# scripts/gpu_c5.py.   bash scripts/pmc_sweep256.sh
set -o pipefail
export TMPDIR=/tmp BL_N=${BL_N:-12500000}
mkdir -p gpurun_out
rm -rf gpurun_out/s256_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s256_stats -- python3 scripts/gpu_c5.py > gpurun_out/s256_stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/s256_$tag -- python3 scripts/gpu_c5.py > gpurun_out/s256_$tag.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/s256_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("sweep", "reduce_256", "k_beta", "xwx", "psi_omega")):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg_ms {float(r["AverageNs"]) / 1e6:8.3f} min {float(r["MinNs"]) / 1e6:8.3f} max {float(r["MaxNs"]) / 1e6:8.3f}')
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    f = glob.glob(f"gpurun_out/s256_{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("sweep", "reduce_256")):
            acc[r["Kernel_Name"].split("(")[1 if r["Kernel_Name"].startswith("(") else 0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for cn, v in cs.items():
            m = sum(v) / len(v)
            extra = f" = {2 * m * 1024 / 1e9:.2f} GB (x2)" if cn == "FETCH_SIZE" else f" = {m * 1024 / 1e9:.2f} GB" if cn == "WRITE_SIZE" else ""
            print(f"{k:40s} {cn:26s} {m:.5g}{extra}")
PY
