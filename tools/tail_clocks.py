"""Where a round of the persistent tail kernel spends its time (needs tools/ab/tail_clocks.so: tools/build_variant.sh tail_clocks
-DSC_TAIL_CLOCKS, SC_LIB_PATH set to it).  Whole Fiat-Shamir proofs of config 3's shape at nv (default 12); block 0's 100 MHz stamps per
tail round: challenge known -> tables bound -> node sums ready -> message published -> next challenge fetched."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shapes, nt = ([[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10) if os.environ.get("SC_SHAPE", "c3") == "c3" else ([[0, 1]], 2)
tabs = [cref.synth_table(2024, s, 1 << nv) for s in range(nt)]
coefs = cref.synth_table(2024, 1000, len(shapes))
poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
lib = C.CDLL(os.environ["SC_LIB_PATH"])
clk = (C.c_uint64 * 512)()
for rep in range(4):
    st.reset()
    st.prove()
assert lib.sc_debug_tail_clocks(clk) == 0
c = np.array(list(clk), dtype=np.int64).reshape(64, 8)
n_tail = min(nv, 12)
print(f"nv={nv} shape={shapes}: tail rounds (pairs from {1 << (n_tail - 1)} down), us per phase")
print("  j  pairs   bind+barrier   sums+barrier   finalize+publish   host round trip   total")
for j in range(n_tail):
    r = c[j]
    ph = [(r[1] - r[0]) / 100.0, (r[2] - r[1]) / 100.0, (r[3] - r[2]) / 100.0, (r[4] - r[3]) / 100.0 if j + 1 < n_tail else 0.0]
    print(f" {j:2d} {1 << (n_tail - 1 - j):6d}   {ph[0]:10.2f}   {ph[1]:12.2f}   {ph[2]:14.2f}   {ph[3]:14.2f}   {sum(ph):7.2f}")
