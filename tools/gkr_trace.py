"""SC_GKR_TRACE=1 python tools/gkr_trace.py: stage times of one config-5 GKR proof (dim 20) on stderr."""
import sys,os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sumcheck_amd as sc
from oracle import cref
dim=20; n=1<<dim; rng=np.random.default_rng(1)
idx=np.unique(rng.integers(0,1<<(3*dim),size=2*n,dtype=np.uint64))[:n]
vals,f2,f3,g=cref.synth_table(1,1,n),cref.synth_table(1,2,n),cref.synth_table(1,3,n),cref.synth_table(1,4,dim)
f1=sc.SparseMultilinearExtension(3*dim,idx,vals); m2=sc.DenseMultilinearExtension(dim,f2); m3=sc.DenseMultilinearExtension(dim,f3)
for i in range(3):
    if i==2: os.environ['SC_GKR_TRACE']='1'
    sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(),f1,m2,m3,g)
