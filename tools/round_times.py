#!/usr/bin/env python3
"""Per-round device time (HIP events around each round's kernels, no profiler) and wall time: tools/round_times.py [nv] [c3|c4]
(c3: config 3's four products over ten tables, default; c4: config 4's one product of three; gkr: one product of two, a GKR phase)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sumcheck_amd as sc
from sumcheck_amd import _lib
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
shapes = {"c4": [[0, 1, 2]], "gkr": [[0, 1]]}.get(sys.argv[2] if len(sys.argv) > 2 else "c3", [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]])
NT = 1 + max(max(s) for s in shapes)
dev = torch.device("cuda:0")
tabs = []
for u in range(NT):
    t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(0x5C20241008, u, 0, 1 << nv, C.c_void_p(t.data_ptr())))
    tabs.append(t)
ct = torch.empty((len(shapes), 4), dtype=torch.int64, device=dev)
_lib.check(sc.lib().sc_synth_table_device(0x5C20241008, 1000, 0, len(shapes), C.c_void_p(ct.data_ptr())))
coefs = ct.cpu().numpy().view(np.uint64)
mles = [sc.DenseMultilinearExtension(nv, t) for t in tabs]
poly = sc.ListOfProductsOfPolynomials(nv)
for k, sh in enumerate(shapes):
    poly.add_product([mles[i] for i in sh], coefs[k])
st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
st.set_timing(True)
rng = sc.Blake2b512Rng.setup()
for rep in range(3):
    st.reset()
    v = None
    dev_ms, wall = [], []
    for i in range(nv):
        t0 = time.perf_counter()
        m = sc.IPForMLSumcheck.prove_round(st, v)
        wall.append((time.perf_counter() - t0) * 1e3)
        dev_ms.append(st.last_round_ms())
        rng.feed(m)
        v = sc.IPForMLSumcheck.sample_round(rng)
print("round  device_ms  wall_ms")
for i in range(nv):
    print(f"{i+1:3d}  {dev_ms[i]:8.3f}  {wall[i]:8.3f}")
print("sum device", sum(dev_ms), "sum wall", sum(wall))
