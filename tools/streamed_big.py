"""Out-of-core at the size SURVEY 8(f4) names (nv >= 30): three tables of 2^nv entries (nv = 30: 96 GiB) kept in PINNED host memory, proved by
a streamed handle (sc_prover_init_streamed: rounds 1 and 2 pull the tables through HBM chunk by chunk, PCIe-bound) and -- for the check --
compared message by message with the CPU oracle's proof of the same tables (about a minute of CPU at nv = 30 under the box's 16-core quota).
python tools/streamed_big.py [nv=30] [check=1]   -> one JSON object on stdout.  Host memory: 3 x 2^nv x 32 B pinned + the oracle's copies."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 30
check = (sys.argv[2] if len(sys.argv) > 2 else "1") != "0"
SEED, shapes, nt = 0x5C20241008, [[0, 1, 2]], 3
dev = torch.device("cuda:0")
host, n = [], 1 << nv
piece = min(n, 1 << 26)  # generate on the device in pieces of 2 GiB, land them in pinned memory
for u in range(nt):
    h = torch.empty((n, 4), dtype=torch.int64, pin_memory=True)
    t = torch.empty((piece, 4), dtype=torch.int64, device=dev)
    for first in range(0, n, piece):
        _lib.check(sc.lib().sc_synth_table_device(SEED, u, first, piece, C.c_void_p(t.data_ptr())))
        h[first:first + piece].copy_(t)
    host.append(h)
    del t
torch.cuda.empty_cache()
coefs = cref.synth_table(SEED, 1000, len(shapes))
mles = [sc.DenseMultilinearExtension(nv, h) for h in host]
poly = sc.ListOfProductsOfPolynomials(nv)
for k, sh in enumerate(shapes):
    poly.add_product([mles[i] for i in sh], coefs[k])
st = sc.IPForMLSumcheck.prover_init(poly, streamed_chunk_log2=0)
ts, proof = [], None
for i in range(3):
    st.reset()
    t0 = time.perf_counter(); proof = np.asarray(st.prove()); ts.append(time.perf_counter() - t0)
out = {"nv": nv, "shapes": shapes, "tables_GiB": nt * n * 32 / 2**30, "streamed_ms": [round(1e3 * t, 1) for t in ts],
       "pcie_GBps": 2 * nt * n * 32 / min(ts) / 1e9, "hbm_free_GiB_during": torch.cuda.mem_get_info()[0] / 2**30}
if check:
    t0 = time.perf_counter()
    d = cref.PolyDesc(nv, [(coefs[0], [0, 1, 2])], [h.numpy().view(np.uint64) for h in host])
    want, _ = cref.ml_prove(d, threads=cref.max_threads())
    out["oracle_s"] = round(time.perf_counter() - t0, 1)
    out["rounds_equal"] = int(sum(bool(np.array_equal(proof.reshape(want.shape)[i], want[i])) for i in range(nv)))
    out["ok"] = out["rounds_equal"] == nv
print(json.dumps(out))
