// kernel_common.hpp -- device helpers shared by kernels.hip (latency-bound rounds, finalize, tail) and kernels_big.hip (big rounds).
#pragma once
#include "fr_device.hpp"
#include "fe_device.hpp"
#include "kernels.h"

namespace scd {

__device__ __forceinline__ Fr fr_from_host(const FrHost &h) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.v[2 * i] = (uint32_t)h.l[i];
        r.v[2 * i + 1] = (uint32_t)(h.l[i] >> 32);
    }
    return r;
}

__device__ __forceinline__ FrU fru_from_host(const FrHost &h) {
    FrU r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.v[2 * i] = (uint32_t)h.l[i];
        r.v[2 * i + 1] = (uint32_t)(h.l[i] >> 32);
    }
    return r;
}

// Montgomery form of a (small, signed) evaluation node
__device__ __forceinline__ Fr node_constant(int32_t nv) {
    if (nv == kNodeInf) return fr_zero();
    return nv >= 0 ? fr_from_u32((uint32_t)nv) : fr_neg(fr_from_u32((uint32_t)(-nv)));
}

__device__ __forceinline__ Fr fr_shfl_down(const Fr &a, int off) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = __shfl_down(a.v[i], off, 64);
    return r;
}

// Sum `acc` over the 256 threads of the block; thread 0 returns the total (others: partial garbage).
__device__ __forceinline__ Fr block_sum(Fr acc, uint32_t (*sm)[8]) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = fr_add(acc, fr_shfl_down(acc, off));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads(); // sm reuse across calls
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sm[wave][i] = acc.v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) {
            Fr o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = sm[w][i];
            acc = fr_add(acc, o);
        }
    }
    return acc;
}

// the line through (lo, hi) at node nv (0, 1, inf, -1, 2, -2, 3, ...), limbs re-tightened to [-4, 2^29 + 4): fits either operand of fe_mul
__device__ __forceinline__ Fe fe_line(const Fe &lo, const Fe &hi, const int32_t nv) {
    if (nv == 0) return lo;
    if (nv == 1) return hi;
    const Fe step = fe_sub(hi, lo); // limbs in (-2^29, 2^29)
    if (nv == kNodeInf) return step;
    Fe cur;
    if (nv < 0) {
        cur = fe_sub(lo, step); // -1: 2 lo - hi, within 2^30
        for (int32_t k = -1; k > nv; --k) cur = fe_sub(fe_carry_pass(cur), step);
    } else {
        cur = fe_add(hi, step); // 2: 2 hi - lo
        for (int32_t k = 2; k < nv; ++k) cur = fe_add(fe_carry_pass(cur), step);
    }
    return fe_carry_pass(cur);
}

} // namespace scd
