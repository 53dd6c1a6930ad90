"""Collapse rocprofv3 counter_collection CSVs (one row per dispatch and counter) into per-kernel means.

    python3 scripts/summarize_pmc.py TAG      # text to stdout + gpurun_out/pmc_summary_TAG.json

The JSON holds, per kernel and per launch:
  hbm_bytes_per_launch   = 2 x FETCH_SIZE + WRITE_SIZE.  rocprofv3 reports both in KiB; on gfx950
                           FETCH_SIZE counts half the bytes of a 16-B/lane coalesced streaming read
                           (MI355X_MICROARCH.md, HBM section), hence the factor 2.
  valu_insts_per_launch  = SQ_INSTS_VALU (wave-level instructions)
  gpu_cycles             = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
  valu_busy_frac         = SQ_ACTIVE_INST_VALU / (256 CUs x gpu_cycles): share of CU-cycles with a
                           VALU instruction executing
  valu_issue_frac_min    = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x gpu_cycles): issue-slot use if
                           every instruction were full rate (fp64 transcendental-seed and 32-bit
                           integer multiplies are slower, so this is a lower bound)
and, from the "stall" pass (what the waves do in the cycles no VALU instruction of theirs executes;
MI355X_MICROARCH.md, rocprofv3 PMC slots: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES):
  wait_any_frac          = SQ_WAIT_ANY / SQ_WAVE_CYCLES        wave parked on s_waitcnt / a barrier
  wait_inst_frac         = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   issue stall (dependency / pipe busy)
  active_inst_frac       = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES an instruction of the wave executing
  salu_per_valu          = SQ_INSTS_SALU / SQ_INSTS_VALU
  lds_conflict_frac      = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
and, from the "mfma" pass:
  mfma_insts_per_launch  = SQ_INSTS_MFMA (wave-level matrix instructions)
  mfma_busy_frac         = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x gpu cycles of that pass, SQ_BUSY_CYCLES / 32): the counter
                           adds 64 cycles per v_mfma_f64_16x16x4_f64 and 16 per v_mfma_f64_4x4x4_4b_f64 (nominal pass counts:
                           the measured issue intervals are 101.7 and 12.4 cycles, scripts/experiments/mfma_f64_shapes.hip)
  lds_insts_per_launch   = SQ_INSTS_LDS
and, from the "lanes" pass (r03):
  active_lane_frac       = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): the share of the 64 lanes that are enabled, averaged
                           over the cycles a vector instruction executes (rocprofiler's VALUUtilization / 100): 1 - this is what
                           divergence (exec masks of the queue's refill path, class branches, rare paths) costs
"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
KEEP = ("k_rpg", "k_psi", "k_xwx", "k_beta", "k_reduce", "k_sweep")
SIMDS = 256 * 4
summary = collections.defaultdict(dict)
for kind in ("fetch", "write", "valu", "stall", "mfma", "lanes"):
    files = glob.glob(f"gpurun_out/pmc_{kind}_{tag}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(kind, "no counter file")
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")
            acc[name][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
    print(f"== {kind}: {files[0]}")
    for name, cs in sorted(acc.items()):
        if not any(k in name for k in KEEP):
            continue
        means = {c: sum(v) / len(v) for c, v in cs.items()}
        print(" ", name[:90], {c: (m, len(cs[c])) for c, m in means.items()})
        for c, m in means.items():
            summary[name].setdefault(c, m)        # SQ_INSTS_VALU / SQ_WAVE_CYCLES are in two passes: keep the first

out = {}
for name, m in summary.items():
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").strip()
    short = short[:short.index("(")] if "(" in short else short
    e = {}
    if "FETCH_SIZE" in m:
        e["fetch_bytes_corrected"] = 2.0 * m["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in m:
        e["write_bytes"] = m["WRITE_SIZE"] * 1024.0
    if "fetch_bytes_corrected" in e and "write_bytes" in e:
        e["hbm_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    if "SQ_INSTS_VALU" in m:
        e["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]
        if m.get("GRBM_GUI_ACTIVE"):
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            e["gpu_cycles"] = cyc
            e["valu_issue_frac_min"] = m["SQ_INSTS_VALU"] * 4.0 / (SIMDS * cyc)
            if m.get("SQ_ACTIVE_INST_VALU"):
                e["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] / (256.0 * cyc)
        if m.get("SQ_WAVES"):
            e["waves"] = m["SQ_WAVES"]
    if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in m:
        wc = m["SQ_WAVE_CYCLES"]
        e["wait_any_frac"] = m["SQ_WAIT_ANY"] / wc
        e["wait_inst_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
        e["active_inst_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        if m.get("SQ_INSTS_VALU"):
            e["salu_per_valu"] = m.get("SQ_INSTS_SALU", 0.0) / m["SQ_INSTS_VALU"]
        if m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_THREAD_CYCLES_VALU") and m.get("SQ_ACTIVE_INST_VALU"):
        e["active_lane_frac"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
    if "SQ_INSTS_MFMA" in m:
        e["mfma_insts_per_launch"] = m["SQ_INSTS_MFMA"]
        e["lds_insts_per_launch"] = m.get("SQ_INSTS_LDS", 0.0)
        if m.get("SQ_BUSY_CYCLES"):
            e["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * m["SQ_BUSY_CYCLES"] / 32.0)
    out[short] = e
with open(f"gpurun_out/pmc_summary_{tag}.json", "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
