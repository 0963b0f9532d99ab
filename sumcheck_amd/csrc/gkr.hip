// gkr.hip -- GKR round sumcheck initialisation on the GPU (placeholder until the sparse kernels land).
#include <hip/hip_runtime.h>
#include "../../include/sumcheck_hip.h"
extern "C" int sc_gkr_phase_one(const uint64_t *, const uint64_t *, uint64_t, uint32_t, const uint64_t *, const uint64_t *, uint64_t *,
                                uint64_t *, uint64_t *, uint64_t *) { return SC_ERR_BAD_ARG; }
extern "C" int sc_gkr_phase_two(const uint64_t *, const uint64_t *, uint64_t, uint32_t, const uint64_t *, uint64_t *) { return SC_ERR_BAD_ARG; }
extern "C" int sc_gkr_prove(sc_rng *, const uint64_t *, const uint64_t *, uint64_t, uint32_t, const uint64_t *, const uint64_t *,
                            const uint64_t *, uint64_t *, uint64_t *) { return SC_ERR_BAD_ARG; }
