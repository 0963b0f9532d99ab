import ctypes as C, sys
sys.path.insert(0,'/root/repo')
import sumcheck_amd as sc
from sumcheck_amd import _lib
for variant in (2,5,6):
    for reps in (500,2000,8000):
        ms=C.c_float(); chk=C.c_uint64()
        _lib.check(sc.lib().sc_bench_modmul(524288, reps, variant, C.byref(ms), C.byref(chk)))
        print(variant, reps, round(ms.value,3), hex(chk.value), f"{524288*4*reps/ms.value/1e6:.1f} G/s")
