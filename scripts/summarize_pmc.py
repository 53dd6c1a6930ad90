"""Collapse rocprofv3 counter_collection CSVs (one row per dispatch and counter) into per-kernel means."""
import csv, glob, sys, collections
tag = sys.argv[1]
for kind in ("fetch", "write", "valu"):
    files = glob.glob(f"gpurun_out/pmc_{kind}_{tag}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(kind, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")[:90]
            acc[name][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
    print(f"== {kind}: {files[0]}")
    for name, cs in sorted(acc.items()):
        if not any(k in name for k in ("k_rpg", "k_psi", "k_xwx", "k_beta", "k_reduce")):
            continue
        print(" ", name, {c: (sum(v) / len(v), len(v)) for c, v in cs.items()})
