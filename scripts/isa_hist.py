#!/usr/bin/env python3
"""Instruction histogram of one kernel of a translation unit (gfx950 ISA as hipcc emits it):
    python scripts/isa_hist.py kernels_pg k_rpg_devroye [top]
Static counts over the whole kernel (both sides of every branch).  IEEE division shows up as v_div_scale/v_div_fmas/v_div_fixup,
libm's log/exp as v_frexp/v_ldexp chains: the markers that led to the short forms of bl_fastmath.hpp."""
import collections
import re
import subprocess
import sys

tu, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
licm_off = tu in ("kernels_tasks", "kernels_pg", "kernels_beta", "kernels_sweep1", "kernels_sweep256")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
       f"bayeslogit_amd/csrc/{tu}.hip", "-o", "-"] + (["-mllvm", "-disable-machine-licm"] if licm_off else [])
asm = subprocess.run(cmd, capture_output=True, text=True).stdout.splitlines()
inside, c = False, collections.Counter()
for l in asm:
    if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", l):
        inside = True
        continue
    if inside and re.match(r"^\s+s_endpgm", l):
        break
    if inside:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m:
            c[re.sub(r"_e32|_e64|_dpp|_sdwa", "", m.group(1))] += 1
valu = sum(n for k, n in c.items() if k.startswith("v_") and "mfma" not in k)
print(f"{kern}: VALU {valu}  MFMA {sum(n for k, n in c.items() if 'mfma' in k)}  SALU {sum(n for k, n in c.items() if k.startswith('s_'))}  "
      f"LDS {sum(n for k, n in c.items() if k.startswith('ds_'))}  VMEM {sum(n for k, n in c.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')))}")
for k, n in c.most_common(top):
    print(f"  {k:28s}{n}")
