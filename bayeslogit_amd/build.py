"""Build libbayeslogit_hip.so in-tree with hipcc for gfx950.

    python -m bayeslogit_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libbayeslogit_hip.so")
SOURCES = ["host_state.hip", "kernels_pg.hip", "kernels_tasks.hip", "kernels_gibbs.hip", "kernels_sweep1.hip", "kernels_sweep256.hip", "kernels_xwx4.hip", "kernels_beta.hip", "capi_gibbs.hip", "combine.hip"]
# per-file flags (see the head of the file named)
EXTRA = {"kernels_tasks.hip": ["-mllvm", "-disable-machine-licm"], "kernels_pg.hip": ["-mllvm", "-disable-machine-licm"],
         # (kernels_beta: the 16-double register vectors of the one-wavefront dense routines are allocas until the AMDGPU pass that
         # turns allocas into vectors, whose default budget is a quarter of the registers: the rest would go to scratch)
         "kernels_beta.hip": ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-promote-alloca-to-vector-vgpr-ratio=1"],
         "kernels_sweep1.hip": ["-mllvm", "-disable-machine-licm"],
         "kernels_sweep256.hip": ["-mllvm", "-disable-machine-licm"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    inc = os.path.join(os.path.dirname(HERE), "include")
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return hdrs


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + _deps()):
        subprocess.check_call([HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", path, "-o", obj])
    return obj


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _stale(LIB, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
