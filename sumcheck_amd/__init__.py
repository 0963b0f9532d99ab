"""sumcheck_amd -- MI355X-native prover hot path of the multilinear sumcheck (arkworks-rs/sumcheck API shape).

The compute lives in libsumcheck_hip.so (hand-written HIP for gfx950, C ABI in include/sumcheck_hip.h);
this package is the thin host-side mirror of the reference's public interface.
"""
from ._lib import SumcheckError, lib  # noqa: F401
from .ml_sumcheck import (Blake2b512Rng, DenseMultilinearExtension, IPForMLSumcheck, ListOfProductsOfPolynomials,  # noqa: F401
                          MLSumcheck, PolynomialInfo, ProverMsg, ProverState, SubClaim, VerifierMsg, interpolate_uni_poly)
from .gkr_round_sumcheck import (GKRProof, GKRRoundSumcheck, GKRRoundSumcheckSubClaim, SparseMultilinearExtension,  # noqa: F401
                                 initialize_phase_one, initialize_phase_two, start_phase1_sumcheck, start_phase2_sumcheck)
