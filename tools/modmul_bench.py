#!/usr/bin/env python3
"""Montgomery-multiplication ceiling of one MI355X: dependent fr_mul chains, 4 per lane, all CUs busy.
Prints G Fr-mul/s (variant 0) and G Fr-add/s (variant 1)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sumcheck_amd as sc
from sumcheck_amd import _lib

for variant, name in ((0, "fr_mul_cios"), (2, "fr_mul_comba"), (4, "fr_mul2_comba"), (3, "fr_mul_comba_uniform"), (1, "fr_add")):
    for blocks_per_cu in (2, 4, 8, 16):
        n_threads = 256 * 256 * blocks_per_cu
        reps = 2000 if variant != 1 else 20000
        ms = C.c_float()
        chk = C.c_uint64()
        _lib.check(sc.lib().sc_bench_modmul(n_threads, reps, variant, C.byref(ms), C.byref(chk)))
        ops = n_threads * 4 * reps
        print(f"{name}: threads={n_threads} reps={reps} {ms.value:.3f} ms -> {ops / ms.value / 1e6:.1f} G op/s")
