"""Where k_finalize's time goes (needs tools/ab/fin_clocks.so: tools/build_variant.sh fin_clocks -DSC_FIN_CLOCKS, SC_LIB_PATH set to it).
Runs config 3 at nv=24 round by round with synchronous rounds (policy "pipeline" = 0) and prints the phase stamps of each round's finalize."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sumcheck_amd as sc
from sumcheck_amd import _lib
_lib.set_policy("pipeline", 0)
from oracle import cref
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
shapes, nt = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
dev = torch.device("cuda:0")
mles = []
for s in range(nt):
    t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(77, s, 0, 1 << nv, C.c_void_p(t.data_ptr())))
    mles.append(sc.DenseMultilinearExtension(nv, t))
coefs = cref.synth_table(77, 1000, len(shapes))
poly = sc.ListOfProductsOfPolynomials(nv)
for k, sh in enumerate(shapes):
    poly.add_product([mles[i] for i in sh], coefs[k])
lib = C.CDLL(os.environ["SC_LIB_PATH"])
clk = (C.c_uint64 * 12)()
chal = cref.synth_table(77, 2000, nv)
for rep in range(2):
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    v = None
    for i in range(min(nv, 12)):
        sc.IPForMLSumcheck.prove_round(st, v)
        v = sc.VerifierMsg(chal[i])
        assert lib.sc_debug_fin_clocks(clk) == 0
        c = [int(x) for x in clk]
        d = [(c[j + 1] - c[j]) / 100.0 for j in range(5)]  # 100 MHz -> us
        if rep:
            e = [(c[j] - c[0]) / 100.0 for j in (6, 7, 8, 9, 1)]
            print(f"   sums detail (us since start): shape known {e[0]:.2f}, loads added {e[1]:.2f}, shuffled {e[2]:.2f}, stored {e[3]:.2f}, barrier passed {e[4]:.2f}")
            print(f"round {i+1:2d}: sums {d[0]:6.2f}  products {d[1]:6.2f}  add {d[2]:6.2f}  over-products+store {d[3]:6.2f}  publish {d[4]:6.2f}  total {sum(d):6.2f} us")
