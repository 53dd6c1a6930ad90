"""Event-timed rates of the two fp64 matrix instructions (register-only loops of the library's diagnostic kernels),
1 / 2 / 4 / 8 waves per SIMD -- the host-clock check of scripts/experiments/mfma_f64_shapes (in-kernel cycle counts)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bayeslogit_amd import device as D
for small in (False, True):
    for wv in (1, 2, 4, 8):
        tf = D.mfma_f64_sustained_tflops(wv, iters=20000, small=small)
        print(f"{'v_mfma_f64_4x4x4_4b_f64 (32 accumulators)' if small else 'v_mfma_f64_16x16x4_f64 (10 accumulators)'}  waves/SIMD {wv}: {tf:6.1f} TFLOP/s", flush=True)
