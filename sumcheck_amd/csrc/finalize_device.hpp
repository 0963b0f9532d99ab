// finalize_device.hpp -- the finalize step as device templates: per-block partials of every product -> the round's ProverMsg.
// Used by k_finalize / k_finalize_mb / k_tail_rounds (kernels.hip) and, in the experiments build, by k_round_tree's in-kernel finalize.
#pragma once
#include "kernel_common.hpp"

namespace scd {
// ------------------------------------------------------------------------------------------------
// Finalize: per-block partials of every product -> the round's ProverMsg (D = deg+1 evaluations).
//   phase 1  S_k[t] = sum over blocks of partial_k[blk][t]                       (t <= M_k)
//   phase 2  P_k(t) for the message points t = 0..D-1 = sum_s (c_k W_k)[t][s] S_k[s]   (host-computed Lagrange weights)
//   phase 3  out[t] = sum_k P_k(t)
// One block of 1024 threads; everything here is O(K*D) field operations and latency-bound, so the intermediate vectors
// live in LDS when they fit (kLds; K*D*(D+2) elements) and phase 1 keeps eight partial loads in flight per lane.
// ------------------------------------------------------------------------------------------------
#ifndef FIN_STAMP // (kernels.hip defines it for a -DSC_FIN_CLOCKS build)
#define FIN_STAMP(i)
#endif
constexpr int kFinBlock = 1024; // 16 wavefronts: one per (product, point) combination for typical shapes
constexpr size_t kFinLdsMax = 48 * 1024;
// the body, for a block of BLOCK threads (k_finalize: 1024; the persistent tail kernel: its own block size); `scratch` holds
// K * D * (D + 2) elements (LDS when it fits); prod_of(k) returns the k-th FinProd
// phase 1: S_k[t] = sum over blocks of partial_k[t][blk] -> scratch[k * D + t]
// Everything here is load latency (the partials were written by other XCDs: every dependent load is a trip to memory), so the
// loads of one combination are issued together and the combinations are spread over as many lanes as the block has:
//   nblocks <= 8   eight lanes per (product, node), one partial each, three shuffle steps;
//   otherwise      the v-th VALID (product, node) pair -- t <= M_k, enumerated without the gaps of the K x D grid, so that a
//                  shape with 14 pairs keeps 14 of the 16 wavefronts busy once instead of 16 and then 4 -- gets W = 1, 2 or 4
//                  wavefronts, each lane up to 12 loads in flight; the W wavefront sums meet in LDS.
template <int BLOCK, typename ProdFn>
__device__ __forceinline__ void finalize_sums(const ProdFn &prod_of, const int K, const int D, const int nblocks, const uint4 *__restrict__ partials,
                                              uint4 *__restrict__ scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = BLOCK / 64;
    if (nblocks <= 8) {
        for (int c0 = 0; c0 < K * D; c0 += BLOCK / 8) { // (uniform trip count: every lane takes part in the shuffles)
            const int combo = c0 + (int)(threadIdx.x >> 3), j = threadIdx.x & 7;
            const int k = min(combo, K * D - 1) / D, t = min(combo, K * D - 1) % D;
            const bool live = combo < K * D && t <= (int)prod_of(k).M;
            Fr acc = fr_zero();
            if (live && j < nblocks) acc = fr_load(partials + 2 * (prod_of(k).partial_off + (uint64_t)t * nblocks + j));
            acc = fr_add(acc, fr_shfl_down(acc, 4));
            acc = fr_add(acc, fr_shfl_down(acc, 2));
            acc = fr_add(acc, fr_shfl_down(acc, 1));
            if (live && j == 0) fr_store(scratch + 2 * combo, acc);
        }
        __syncthreads();
        return;
    }
    int n_valid = 0;
    for (int k = 0; k < K; ++k) n_valid += min((int)prod_of(k).M, D - 1) + 1;
    FIN_STAMP(6);
    const int W = n_valid * 4 <= kWaves ? 4 : (n_valid * 2 <= kWaves ? 2 : 1);
    __shared__ uint4 xwave[kWaves * 2];
    for (int v0 = 0; v0 < n_valid; v0 += kWaves / W) {
        const int v = v0 + wave / W, sub = wave % W;
        int k = 0, t = 0;
        bool live = v < n_valid;
        if (live) { // v -> (k, t)
            int rest = v;
            for (k = 0; k < K; ++k) {
                const int cnt = min((int)prod_of(k).M, D - 1) + 1;
                if (rest < cnt) break;
                rest -= cnt;
            }
            t = rest;
        }
        Fr acc = fr_zero();
        if (live) {
            const uint4 *base = partials + 2 * (prod_of(k).partial_off + (uint64_t)t * nblocks);
            constexpr int kLoads = BLOCK >= 1024 ? 12 : 6; // (768 partials = one batch of 12 for k_finalize; the tail kernel keeps its register budget)
            for (int b0 = sub * 64 + lane; b0 < nblocks; b0 += 64 * W * kLoads) {
                Fr x[kLoads];
#pragma unroll
                for (int j = 0; j < kLoads; ++j) {
                    // clamped address + select: the loads are issued back to back (a predicated load would wait for its own data)
                    const int blk = b0 + 64 * W * j;
                    const Fr ld = fr_load(base + 2 * min(blk, nblocks - 1));
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[j].v[i] = blk < nblocks ? ld.v[i] : 0u;
                }
                if constexpr (kLoads == 12) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) x[j] = fr_add(x[j], x[j + 6]);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) x[j] = fr_add(x[j], x[j + 3]);
                acc = fr_add(acc, fr_add(fr_add(x[0], x[1]), x[2]));
            }
        }
        FIN_STAMP(7);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc = fr_add(acc, fr_shfl_down(acc, off));
        FIN_STAMP(8);
        if (W == 1) {
            if (live && lane == 0) fr_store(scratch + 2 * (k * D + t), acc);
        } else {
            if (lane == 0) fr_store(xwave + 2 * wave, acc);
            __syncthreads();
            if (live && lane == 0 && sub == 0) {
                for (int w = 1; w < W; ++w) acc = fr_add(acc, fr_load(xwave + 2 * (wave + w)));
                fr_store(scratch + 2 * (k * D + t), acc);
            }
            __syncthreads(); // xwave is reused by the next pass
        }
    }
    FIN_STAMP(9);
    __syncthreads();
}
// The compact form of phases 2 and 3 (K * D <= 32 and 32 * D <= BLOCK: every shape the benchmarks use): message point t belongs to
// the 32 lanes [32 t, 32 t + 32); lane (k, s) of them holds the single product (c_k W_k)[t][s] * S_k[s], five shuffle steps add them
// up.  No intermediate vector, no barrier between the products and the sums; the weight can be fetched before the node sums exist
// (fin_prefetch_weight at the top of the kernel -- or once per launch in the persistent kernel), which takes its ~1.5 us memory
// latency off the critical path.
template <int BLOCK>
__device__ __forceinline__ bool fin_compact(const int K, const int D) { return K * D <= 32 && 32 * D <= BLOCK; }
template <typename ProdFn>
__device__ __forceinline__ Fr fin_prefetch_weight(const ProdFn &prod_of, const uint4 *__restrict__ Wm, const int K, const int D, const int scaled) {
    const int t = threadIdx.x >> 5, slot = threadIdx.x & 31, k = slot / D, sN = slot % D;
    if (t >= D || k >= K) return fr_zero();
    const int M = (int)prod_of(k).M;
    if (sN > M) return fr_zero();
    const uint64_t woff = prod_of(k).w_off + (((scaled & 1) && (M <= kMaxFusedM || (scaled & 4))) ? (uint64_t)D * (M + 1) : 0);
    return fr_load(Wm + 2 * (woff + (uint64_t)t * (M + 1) + sN));
}
// The message is in host-mapped memory (the stores above); raise its sequence flag behind it: every writing wave drains its stores (they
// have reached the L2 / the fabric once acknowledged), the block meets, ONE lane raises the flag with system-scope release semantics -- its
// write-back covers the whole L2.  Until round 5 every thread fenced to system scope first (a second write-back + invalidate per round:
// ~0.5 us of every round of every proof, profiles/r5i_publish_fence_ab.txt; -DSC_PUBLISH_MODE=0 builds that form).  A RELAXED flag store
// behind the drained stores is NOT enough: the message is cached in L2, the host saw flags before their data (parity failure, same file).
#ifndef SC_PUBLISH_MODE
#define SC_PUBLISH_MODE 1
#endif
__device__ __forceinline__ void fin_publish_flag(uint32_t *__restrict__ h_flag, const uint32_t seq) {
#if SC_PUBLISH_MODE == 0
    __threadfence_system();
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// phases 2 and 3: the node sums in scratch[k * D + t] -> the round message.  w_pre: this thread's fin_prefetch_weight, or null
template <int BLOCK, typename ProdFn>
__device__ __forceinline__ void finalize_message(const ProdFn &prod_of, const uint4 *__restrict__ Wm, const int K, const int D, uint4 *__restrict__ scratch,
                                                 uint4 *__restrict__ out, uint64_t *__restrict__ out_wide, uint4 *__restrict__ h_out,
                                                 uint32_t *__restrict__ h_flag, const uint32_t seq, const int scaled, const Fr *w_pre = nullptr) {
    // `scaled`: bit 0 = the partials carry 2^(-5(M-1)) (second copy of the weights) -- those of products of up to kMaxFusedM multiplicands, with
    // bit 2 those of longer products too (k_tail_slices<kMaxWideM>); bit 1 = tag the widened lanes with this round's
    // sequence number (kernels.h: wide_tag_of) -- an RCCL all-reduce then lands them in host-mapped memory as self-validating words
    const uint64_t wtag = (scaled & 2) ? ((uint64_t)wide_tag_of(seq) << kWideTagShift) : 0;
    if (fin_compact<BLOCK>(K, D)) {
        if ((int)(threadIdx.x & ~63u) < 32 * D) { // whole wavefronts
            const int t = threadIdx.x >> 5, slot = threadIdx.x & 31, k = slot / D, sN = slot % D;
            const bool live = t < D && k < K && sN <= (int)prod_of(k).M;
            const Fr w = w_pre ? *w_pre : fin_prefetch_weight(prod_of, Wm, K, D, scaled);
            Fr acc = fr_zero();
            if (live) acc = fr_mul(w, fr_load(scratch + 2 * (k * D + sN)));
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) acc = fr_add(acc, fr_shfl_down(acc, off));
            if (slot == 0 && t < D) {
                if (out) fr_store(out + 2 * t, acc);
                if (h_out) fr_store(h_out + 2 * t, acc); // host-mapped pinned memory: the message lands on the host without a copy
                if (out_wide) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) out_wide[8 * t + i] = (uint64_t)acc.v[i] | wtag;
                }
            }
        }
        FIN_STAMP(4);
        if (h_flag) fin_publish_flag(h_flag, seq);
        return;
    }
    // phase 2: message point t of product k = sum_s (c_k W_k)[t][s] * S_k[s].  One thread per (k, t, s) does the single
    // Montgomery product (a lone lane needs ~1 us per product, so the M+1 products of a point must not be chained) ...
    for (int idx = threadIdx.x; idx < K * D * D; idx += BLOCK) {
        const int k = idx / (D * D), t = (idx / D) % D, sN = idx % D;
        const int M = (int)prod_of(k).M;
        if (sN > M) continue;
        // partials from the 2^261-radix kernels carry 2^(-5(M-1)); the second copy of the matrix undoes it
        const uint64_t woff = prod_of(k).w_off + (((scaled & 1) && (M <= kMaxFusedM || (scaled & 4))) ? (uint64_t)D * (M + 1) : 0);
        const uint4 *Wk = Wm + 2 * (woff + (uint64_t)t * (M + 1));
        fr_store(scratch + 2 * ((2 * K) * D + idx), fr_mul(fr_load(Wk + 2 * sN), fr_load(scratch + 2 * (k * D + sN))));
    }
    __syncthreads();
    FIN_STAMP(2);
    // ... and one thread per (k, t) adds them up
    for (int combo = threadIdx.x; combo < K * D; combo += BLOCK) {
        const int k = combo / D;
        const int M = (int)prod_of(k).M;
        Fr acc = fr_zero();
        for (int sN = 0; sN <= M; ++sN) acc = fr_add(acc, fr_load(scratch + 2 * ((2 * K) * D + combo * D + sN)));
        fr_store(scratch + 2 * ((K + k) * D + combo % D), acc);
    }
    __syncthreads();
    FIN_STAMP(3);
    // phase 3b: sum over products
    for (int t = threadIdx.x; t < D; t += BLOCK) {
        Fr acc = fr_zero();
        for (int k = 0; k < K; ++k) acc = fr_add(acc, fr_load(scratch + 2 * ((K + k) * D + t)));
        if (out) fr_store(out + 2 * t, acc);
        if (h_out) fr_store(h_out + 2 * t, acc); // host-mapped pinned memory: the message lands on the host without a copy
        if (out_wide) {
#pragma unroll
            for (int i = 0; i < 8; ++i) out_wide[8 * t + i] = (uint64_t)acc.v[i] | wtag;
        }
    }
    FIN_STAMP(4);
    if (h_flag) fin_publish_flag(h_flag, seq);
}
template <int BLOCK, typename ProdFn>
__device__ __forceinline__ void finalize_body(const ProdFn &prod_of, const uint4 *__restrict__ Wm, const int K, const int D, const int nblocks,
                                              const uint4 *__restrict__ partials, uint4 *__restrict__ scratch, uint4 *__restrict__ out,
                                              uint64_t *__restrict__ out_wide, uint4 *__restrict__ h_out, uint32_t *__restrict__ h_flag, const uint32_t seq,
                                              const int scaled, const Fr *w_pre = nullptr) {
    FIN_STAMP(0);
    finalize_sums<BLOCK>(prod_of, K, D, nblocks, partials, scratch);
    FIN_STAMP(1);
    finalize_message<BLOCK>(prod_of, Wm, K, D, scratch, out, out_wide, h_out, h_flag, seq, scaled, w_pre);
    FIN_STAMP(5);
}

} // namespace scd
