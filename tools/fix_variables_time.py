"""DenseMultilinearExtension::fix_variables (sc_fix_variables) on a resident 2^nv table: python tools/fix_variables_time.py [nv]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sumcheck_amd as sc
from sumcheck_amd import _lib
from oracle import cref
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t = torch.empty((1 << nv, 4), dtype=torch.int64, device="cuda:0")
_lib.check(sc.lib().sc_synth_table_device(77, 0, 0, 1 << nv, C.c_void_p(t.data_ptr())))
m = sc.DenseMultilinearExtension(nv, t)
pt = cref.synth_table(77, 2000, nv)
out = []
for k in (1, 3, 6, nv):
    ts = []
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = m.fix_variables(pt[:k])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out.append(f"k={k}: {1e3*np.median(ts[2:]):.3f} ms")
print(f"fix_variables nv={nv}: " + "  ".join(out))
