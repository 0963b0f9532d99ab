#!/usr/bin/env python3
"""tools/e2e_trace_run.py: three staged sc_prover_init + prove over pinned host tables of config 3 (nv = 24), for a rocprofv3
--kernel-trace --memory-copy-trace timeline of the staged initialisation (tools/e2e_phases.py gives the wall times)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sumcheck_amd as sc
from sumcheck_amd import _lib
import ctypes as C
nv = 24
shapes, U = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
dev = torch.device("cuda", 0)
n = 1 << nv
host = []
for u in range(U):
    t = torch.empty((n, 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(0x5C20241008, u, 0, n, C.c_void_p(t.data_ptr())))
    host.append(t.cpu().pin_memory())
    del t
coefs = np.ones((len(shapes), 4), dtype=np.uint64)
coefs[:, 1:] = 0
mles = [sc.DenseMultilinearExtension(nv, h) for h in host]
poly = sc.ListOfProductsOfPolynomials(nv)
for k, sh in enumerate(shapes):
    poly.add_product([mles[i] for i in sh], coefs[k])
for _ in range(3):
    st = sc.IPForMLSumcheck.prover_init(poly)
    st.prove()
    st.close()
