"""Where a round of k_tail_slices (kernels_tail.hip) spends its time: tools/build_variant.sh tail_clocks -DSC_TAIL_CLOCKS, SC_LIB_PATH set to it.
Whole Fiat-Shamir proofs at nv (default 12) of config 3's shape (SC_SHAPE=c3), config 2's (c2) or a GKR phase's (gkr); block 0's 100 MHz
stamps per tail round: start -> challenge in hand -> slice bound -> own sums -> every block's partials in -> message published."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shapes, nt = {"c3": ([[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10), "c2": ([[0, 1, 2]], 3), "gkr": ([[0, 1]], 2)}[os.environ.get("SC_SHAPE", "c3")]
tabs = [cref.synth_table(2024, s, 1 << nv) for s in range(nt)]
coefs = cref.synth_table(2024, 1000, len(shapes))
poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
lib = C.CDLL(os.environ["SC_LIB_PATH"])
clk = (C.c_uint64 * 512)()
for rep in range(4):
    st.reset()
    st.prove()
assert lib.sc_debug_tail_slices_clocks(clk) == 0
c = np.array(list(clk), dtype=np.int64).reshape(64, 8)
n_tail = min(nv, 15)  # the slices tail takes over at 2^14 pairs
print(f"nv={nv} shape={shapes}: k_tail_slices rounds (pairs from {1 << (n_tail - 1)} down), us per phase (block 0)")
print("  j  pairs   challenge     bind    own sums   partials in   finalize+publish   total (start to published)   to next start")
for j in range(n_tail):
    r = c[j]
    ph = [(r[i + 1] - r[i]) / 100.0 for i in range(5)]
    nxt = (c[j + 1][0] - r[5]) / 100.0 if j + 1 < n_tail else 0.0
    extra = f"   [partials: accumulators complete +{(r[6] - r[3]) / 100.0:.2f}, folded +{(r[7] - r[6]) / 100.0:.2f}, added +{(r[4] - r[7]) / 100.0:.2f}]" if r[6] > r[3] else ""
    print(f" {j:2d} {1 << (n_tail - 1 - j):6d}   {ph[0]:8.2f} {ph[1]:8.2f} {ph[2]:10.2f} {ph[3]:12.2f} {ph[4]:16.2f} {(r[5] - r[0]) / 100.0:18.2f} {nxt:20.2f}{extra}")
