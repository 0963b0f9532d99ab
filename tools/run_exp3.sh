SC_WAIT_SPINS=1 SC_PIPELINE=1 SC_HOST_TRACE=1 timeout 120 python - <<'PY' 2>&1 | tail -30
import numpy as np, sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
nv, shapes = 12, [[0, 1, 2], [1]]
tabs = [cref.synth_table(7, s, 1 << nv) for s in range(3)]
coefs = cref.synth_table(7, 1000, len(shapes))
poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs)
try:
    sc.MLSumcheck.prove(poly)
    print("NO-ERROR")
except sc.SumcheckError as e:
    print("ERROR", e)
PY
