#!/usr/bin/env python3
"""Do the caller's tables collide in the memory channels?  Ten 512 MiB tables at power-of-two distances (separate allocations) against the
same tables carved out of one buffer at skewed distances: whole proofs and rounds 1-3.  python tools/skew_probe.py [skew_bytes ...]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sumcheck_amd as sc
from sumcheck_amd import _lib
nv = 24
shapes = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]
dev = torch.device("cuda:0")
ct = torch.empty((4, 4), dtype=torch.int64, device=dev)
_lib.check(sc.lib().sc_synth_table_device(0x5C20241008, 1000, 0, 4, C.c_void_p(ct.data_ptr())))
coefs = ct.cpu().numpy().view(np.uint64)
def run(skew):
    n = 1 << nv
    if skew is None:
        tabs = [torch.empty((n, 4), dtype=torch.int64, device=dev) for _ in range(10)]
    else:
        per = n * 4 + skew // 8
        big = torch.empty((10 * per + 64,), dtype=torch.int64, device=dev)
        tabs = [big[u * per: u * per + n * 4].view(n, 4) for u in range(10)]
    for u, t in enumerate(tabs):
        _lib.check(sc.lib().sc_synth_table_device(0x5C20241008, u, 0, n, C.c_void_p(t.data_ptr())))
    print("skew", skew, "addresses mod 2^20:", [hex(t.data_ptr() & 0xFFFFF) for t in tabs[:4]], "distance", hex(tabs[1].data_ptr() - tabs[0].data_ptr()))
    mles = [sc.DenseMultilinearExtension(nv, t) for t in tabs]
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    for _ in range(3):
        st.reset(); st.prove()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        st.reset(); st.prove()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    st.set_timing(True)
    rng = sc.Blake2b512Rng.setup()
    st.reset(); v = None; r = []
    for i in range(4):
        m = sc.IPForMLSumcheck.prove_round(st, v); r.append(st.last_round_ms()); rng.feed(m); v = sc.IPForMLSumcheck.sample_round(rng)
    m = sc.IPForMLSumcheck.prove_round(st, v); r.append(st.last_round_ms())
    print(f"   proof {ms:.3f} ms; rounds 1-4 (event ms): " + " ".join(f"{x:.3f}" for x in r[1:]))
    st.close()
for s in [None] + [int(a) for a in sys.argv[1:]] + [None]:
    run(s)
