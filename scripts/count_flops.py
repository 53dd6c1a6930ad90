#!/usr/bin/env python3
"""The flops-per-call table behind bench.py's fp64-VALU rate ("achieved fp64 VALU op rate against the
nominal 78.6 TFLOP/s ... costed at a stated flops/call table").

    python scripts/count_flops.py            # -> profiles/flops_table.json   (hipcc only; no GPU)

scripts/experiments/flops_calls.hip holds one tiny kernel per building block of the draw kernels; this script compiles
it for gfx950 with -S and counts, per kernel, the vector instructions and the fp64 flops of the emitted ISA, minus the
load/store harness (kernel `baseline`):
    v_fma_f64 / v_fmac_f64: 2 flops; v_mul_f64 / v_add_f64: 1; v_rcp/rsq/sqrt_f64 (seeds), v_ldexp/frexp/cvt/cmp/cndmask,
    fp32 and integer instructions: 0 flops (they still take a VALU issue slot: counted in `valu`).
Branches are counted as written (both sides of a rare branch: upper bound for the rare paths, e.g. the series walk)."""
import collections
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "experiments", "flops_calls.hip")
FLOPS = {"v_fma_f64": 2, "v_fmac_f64": 2, "v_mul_f64": 1, "v_add_f64": 1, "v_pk_fma_f64": 4, "v_pk_mul_f64": 2, "v_pk_add_f64": 2}


def main():
    asm = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm",
                                   "-disable-machine-licm", "-S", "--cuda-device-only", SRC, "-o", "-"], text=True)
    kern = {}
    cur = None
    for line in asm.splitlines():
        m = re.match(r"^(fc_\w+):", line)
        if m:
            cur = m.group(1)[3:]
            kern[cur] = collections.Counter()
            continue
        if cur and re.match(r"^\s+s_endpgm", line):
            cur = None
            continue
        if cur:
            m = re.match(r"^\s+(v_\w+)", line)
            if m:
                op = re.sub(r"_e32|_e64|_dpp|_sdwa", "", m.group(1))
                kern[cur][op] += 1
    base = kern.pop("baseline")
    table = {}
    for name, c in kern.items():
        c = c - base
        valu = sum(c.values())
        flops = sum(FLOPS.get(op, 0) * n for op, n in c.items())
        f64 = sum(n for op, n in c.items() if op.endswith("_f64"))
        table[name] = {"valu_instructions": valu, "fp64_instructions": f64, "fp64_flops": flops,
                       "top": dict(c.most_common(6))}
    out = {"source": "scripts/count_flops.py over scripts/experiments/flops_calls.hip (gfx950 ISA, -O3, machine-LICM off)",
           "flops_rule": "v_fma_f64 = 2, v_mul_f64 / v_add_f64 = 1, everything else 0", "per_call": table}
    path = os.path.join(ROOT, "profiles", "flops_table.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in table.items():
        print(f"{k:16s} valu {v['valu_instructions']:4d}  fp64 {v['fp64_instructions']:4d}  flops {v['fp64_flops']:4d}")


if __name__ == "__main__":
    sys.exit(main())
