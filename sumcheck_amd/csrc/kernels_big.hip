// kernels_big.hip -- the big-round kernels of the sumcheck prover hot path (rounds with more than kSmallRoundPairs = 2^14 pairs): the fused bind + sum
// over the hypercube (reference src/ml_sumcheck/protocol/prover.rs:84-89 + 110-148) as a static product tree in carry-free arithmetic.
// Its own translation unit so that it compiles beside kernels.hip (both are minutes of hipcc); see kernels.hip for the overview.
#include "kernel_common.hpp"
#include "finalize_device.hpp"
#include "load_factor.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace scd {

#ifdef SC_EXPERIMENTS // cross-check variant: saturated 8 x u32 Comba arithmetic (SC_KERNEL=0 SC_FE=0)
// ------------------------------------------------------------------------------------------------
// K4: fused bind + product-sum for one product with M multiplicands (M <= kMaxFusedM)
// ------------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(kBlock) void k_prod_round(const ProdArgs A, const FrHost r_h, const uint64_t n_pairs,
                                                       uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    const FrU r = fru_from_host(r_h); // the challenge is a kernel argument: its limbs stay in SGPRs
    Fr acc[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) acc[t] = fr_zero();

    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_pairs; b += stride) {
        Fr prod[M + 1];
        bool first = true;
        for (int s = 0; s < A.n_slots; ++s) {
            Fr lo, hi;
            if (A.slot[s].mode == 0) {
                const uint4 *p = A.slot[s].src + 4 * b; // pair b = 64 contiguous bytes
                lo = fr_load(p);
                hi = fr_load(p + 2);
            } else {
                const uint4 *p = A.slot[s].src + 8 * b; // entries 4b..4b+3 = 128 contiguous bytes
                const Fr e0 = fr_load(p), e1 = fr_load(p + 2), e2 = fr_load(p + 4), e3 = fr_load(p + 6);
                Fr ml, mh; // the two binds of a pair are independent: interleave their instruction streams
                fr_mul2_comba_u(fr_sub(e1, e0), r, fr_sub(e3, e2), r, ml, mh);
                lo = fr_add(e0, ml);
                hi = fr_add(e2, mh);
                uint4 *q = A.slot[s].dst + 4 * b;
                fr_store(q, lo);
                fr_store(q + 2, hi);
            }
            const Fr step = fr_sub(hi, lo);
            const uint32_t e = A.slot[s].exp;
            Fr curP = hi, curN = lo; // walking outwards from 1 and 0 along the line
            Fr cur[M + 1];
#pragma unroll
            for (int t = 0; t <= M; ++t) {
                const int32_t nv = node_value(t);
                if (nv == 0) cur[t] = lo;
                else if (nv == 1) cur[t] = hi;
                else if (nv == kNodeInf) cur[t] = step;
                else if (nv < 0) { curN = fr_sub(curN, step); cur[t] = curN; }
                else { curP = fr_add(curP, step); cur[t] = curP; }
            }
            uint32_t k = 0;
            if (first) {
#pragma unroll
                for (int t = 0; t <= M; ++t) prod[t] = cur[t];
                k = 1;
            }
            for (; k < e; ++k) { // nodes in pairs: two independent Montgomery products per asm stream
#pragma unroll
                for (int t = 0; t + 1 <= M; t += 2) fr_mul2_comba(prod[t], cur[t], prod[t + 1], cur[t + 1], prod[t], prod[t + 1]);
                if ((M + 1) % 2 == 1) prod[M] = fr_mul(prod[M], cur[M]);
            }
            first = false;
        }
#pragma unroll
        for (int t = 0; t <= M; ++t) acc[t] = fr_add(acc[t], prod[t]);
    }
#pragma unroll
    for (int t = 0; t <= M; ++t) {
        const Fr s = block_sum(acc[t], sm);
        if (threadIdx.x == 0) fr_store(partials + 2 * ((uint64_t)t * gridDim.x + blockIdx.x), s);
    }
}

#endif // SC_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// K4 in carry-free arithmetic (fe_device.hpp): the same fused bind + product-sum, 9 x 29-bit signed limbs.
// `r32` is the challenge times 2^5 (host), so fe_mul_u(hi - lo, r32) is r*(hi-lo) in the tables' R = 2^256 form;
// the M-1 products of a term leave the factor 2^(-5(M-1)), which k_finalize removes through the scaled coefficient.
// ------------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(kBlock) void k_prod_round_fe(const ProdArgs A, const FrHost r32_h, const uint64_t n_pairs,
                                                          uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    const FeU r = feu_from_host(r32_h);
    Fe acc[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) acc[t] = fe_zero();

    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t iter = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_pairs; b += stride, ++iter) {
        Fe prod[M + 1];
        bool first = true;
        for (int s = 0; s < A.n_slots; ++s) {
            Fe lo, hi;
            if (A.slot[s].mode == 0) {
                const uint4 *p = A.slot[s].src + 4 * b;
                lo = fe_from_fr(fr_load(p));
                hi = fe_from_fr(fr_load(p + 2));
            } else {
                const uint4 *p = A.slot[s].src + 8 * b;
                const Fe e0 = fe_from_fr(fr_load(p)), e1 = fe_from_fr(fr_load(p + 2));
                const Fe e2 = fe_from_fr(fr_load(p + 4)), e3 = fe_from_fr(fr_load(p + 6));
                const Fe l0 = fe_add(e0, fe_mul_u(fe_sub(e1, e0), r));
                const Fe h0 = fe_add(e2, fe_mul_u(fe_sub(e3, e2), r));
                const Fr lc = fe_to_fr(l0), hc = fe_to_fr(h0); // tables stay canonical in the reference layout
                uint4 *q = A.slot[s].dst + 4 * b;
                fr_store(q, lc);
                fr_store(q + 2, hc);
                lo = fe_from_fr(lc);
                hi = fe_from_fr(hc);
            }
            const Fe step = fe_sub(hi, lo); // limbs in (-2^29, 2^29)
            const uint32_t e = A.slot[s].exp;
            Fe curP = hi, curN = lo;
#pragma unroll
            for (int t = 0; t <= M; ++t) {
                const int32_t nv = node_value(t);
                Fe cur;
                if (nv == 0) cur = lo;
                else if (nv == 1) cur = hi;
                else if (nv == kNodeInf) cur = step;
                else if (nv < 0) { // -1 = 2lo - hi is within the 2^30 limb bound as is; further out re-tighten first
                    curN = (nv == -1) ? fe_sub(curN, step) : fe_sub(fe_carry_pass(curN), step);
                    cur = curN;
                } else {
                    curP = (nv == 2) ? fe_add(curP, step) : fe_add(fe_carry_pass(curP), step);
                    cur = curP;
                }
                uint32_t k = 0;
                if (first) { prod[t] = (nv == 0 || nv == 1) ? cur : fe_carry_pass(cur); k = 1; }
                for (; k < e; ++k) prod[t] = fe_mul(cur, prod[t]);
            }
            first = false;
        }
#pragma unroll
        for (int t = 0; t <= M; ++t) acc[t] = fe_carry_pass(fe_add(acc[t], prod[t]));
        if ((iter & 31u) == 31u) { // keep the top limb far from 2^31 on very long grid-stride loops
#pragma unroll
            for (int t = 0; t <= M; ++t) acc[t] = fe_from_fr(fe_to_fr(acc[t]));
        }
    }
#pragma unroll
    for (int t = 0; t <= M; ++t) {
        const Fr s = block_sum(fe_to_fr(acc[t]), sm);
        if (threadIdx.x == 0) fr_store(partials + 2 * ((uint64_t)t * gridDim.x + blockIdx.x), s);
    }
}

// ------------------------------------------------------------------------------------------------
// K4, product tree: the production big-round kernel.  `A` lists exactly M FACTORS (a table that occurs twice in the
// product is listed twice), so the multiplication schedule is static per M:
//   M = 2:  3 products  (nodes 0, 1, inf)
//   M = 3:  q = f0 f1 at {0, 1, inf} (3), extended to -1 by additions, times f2 at {0, 1, inf, -1} (4)      =  7 (not 8)
//   M = 4:  qa = f0 f1, qb = f2 f3 at {0, 1, inf} (6), both extended to {-1, 2}, qa qb at the five nodes (5) = 11 (not 15)
//   (M >= 5 runs node by node in k_prod_round_fe, M-1 products each.)
// A quadratic q with q0 = q(0), q1 = q(1), qi = leading coefficient has q(-1) = 2 qi - q1 + 2 q0 and
// q(2) = 2 qi + 2 q1 - q0: three lazy limb-wise additions and one carry pass in the 29-bit representation.
// Slot modes: 0 read this round's table; 1 bind the previous table, store, use; 3 bind without storing (a repeated
// factor, or a table that an earlier product of the same launch stores).
// ------------------------------------------------------------------------------------------------
// one product's pass of a block over its share of the pairs: row = this block's M+1 partial sums of that product
// kSkip1: node 1's sum is not computed -- the finalize step derives it from the previous round (S(0) + S(1) = that round's
// polynomial at the challenge, product by product: ClaimArgs in kernels.h); binding rounds only
template <int M, bool kR1 = false, bool kChain = kChainDefault, bool kSkip1 = false>
__device__ __forceinline__ void tree_pass(const Slot *S, const int32_t (&r)[kBindLds], const uint64_t n_pairs, uint4 *__restrict__ row, uint32_t (*sm)[8],
                                          int32_t *lacc, const uint32_t part_stride = 0) {
    // (part_stride: blocks per node row of the partial sums when this launch fills only a SECTION of them -- round 1 of a staged
    // sc_prover_init, chunk by chunk under the host-to-device copy; 0: this launch's own grid)
    const uint64_t pstride = part_stride ? part_stride : gridDim.x;
    // The M+1 running sums live in LDS (limb-planar, one column per thread: lacc[(9 t + limb) * kBlock + tid], conflict-free and
    // private to the thread, so no barrier): 45 VGPRs less for M = 4, one more resident block per CU.
    static_assert(!(kSkip1 && kR1), "round 1 has no previous round to take node 1 from");
    int32_t *my = lacc + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 9 * (M + 1); ++i) my[i * kBlock] = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t iter = 0;
    uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    // Two and three multiplicands: TWO pairs per iteration (b and b + stride).  The final products of a node only ever feed that
    // node's running sum, so the two pairs' products share one Montgomery reduction (fe_mul2_sum: 234 multiply-adds instead of 306).
    // The live set of these shapes (at most 2 x 5 elements) fits the 168 registers of three resident blocks with 16 spills.  Four
    // multiplicands stay one pair at a time: twelve quadratic coefficients across two pairs spill 47 registers, and the scratch traffic
    // costs round 2 more (+75 us) than the shared reductions save in round 1 (-23 us) -- measured, same box.
#ifdef SC_M4_PAIRS // A/B build (tools/build_variant.sh): two pairs per iteration for four multiplicands in the binding rounds too
    constexpr bool kM4Pairs = true;
#else
    constexpr bool kM4Pairs = kR1;
#endif
    if constexpr (M == 2 || M == 3 || (M == 4 && kM4Pairs)) {
        for (; b + stride < n_pairs; b += 2 * stride, ++iter) {
            Fe P[M + 1];
            const uint64_t b2 = b + stride;
            [[maybe_unused]] auto accumulate = [&](const int t, const Fe &v) { // node t's running sum += v (the same schedule of carry passes as below)
                Fe acc;
#pragma unroll
                for (int l = 0; l < 9; ++l) acc.l[l] = my[(9 * t + l) * kBlock];
                acc = fe_add(acc, v);
                if (iter & 1u) acc = fe_carry_pass(acc);
                if ((iter & 31u) == 31u) acc = fe_from_fr(fe_to_fr(acc));
#pragma unroll
                for (int l = 0; l < 9; ++l) my[(9 * t + l) * kBlock] = acc.l[l];
            };
            if constexpr (M == 4) { // round 1 only: without the bind path the twelve quadratic coefficients of two pairs fit
                Fe a0, a1, ai, b0, b1, bi, c0, c1, ci, d0, d1, di; // a, b: pair b's two quadratics; c, d: pair b2's
                {
                    Fe l0, h0, l1, h1;
                    LoadFactor<0, kR1, kChain>::run(S, b, r, l0, h0);
                    LoadFactor<1, kR1, kChain>::run(S, b, r, l1, h1);
                    a0 = fe_mul<kChain>(l0, l1);
                    a1 = fe_mul<kChain>(h0, h1);
                    ai = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
                }
                fe_pin3(a0, a1, ai);
                {
                    Fe l2, h2, l3, h3;
                    LoadFactor<2, kR1, kChain>::run(S, b, r, l2, h2);
                    LoadFactor<3, kR1, kChain>::run(S, b, r, l3, h3);
                    b0 = fe_mul<kChain>(l2, l3);
                    b1 = fe_mul<kChain>(h2, h3);
                    bi = fe_mul<kChain>(fe_sub(h2, l2), fe_sub(h3, l3));
                }
                fe_pin3(b0, b1, bi);
                {
                    Fe l0, h0, l1, h1;
                    LoadFactor<0, kR1, kChain>::run(S, b2, r, l0, h0);
                    LoadFactor<1, kR1, kChain>::run(S, b2, r, l1, h1);
                    c0 = fe_mul<kChain>(l0, l1);
                    c1 = fe_mul<kChain>(h0, h1);
                    ci = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
                }
                fe_pin3(c0, c1, ci);
                {
                    Fe l2, h2, l3, h3;
                    LoadFactor<2, kR1, kChain>::run(S, b2, r, l2, h2);
                    LoadFactor<3, kR1, kChain>::run(S, b2, r, l3, h3);
                    d0 = fe_mul<kChain>(l2, l3);
                    d1 = fe_mul<kChain>(h2, h3);
                    di = fe_mul<kChain>(fe_sub(h2, l2), fe_sub(h3, l3));
                }
                // Every node's product goes into its running sum as soon as it exists (nothing waits in registers), and BEFORE nodes -1
                // and 2 each quadratic's three coefficients are replaced by its two extension values -- q(-1) = 2 q(0) + 2 q(inf) - q(1),
                // q(2) = 2 q(1) + 2 q(inf) - q(0), re-tightened for the shared reduction -- so that those products see eight live
                // elements instead of twelve coefficients plus four temporaries (the 136 bytes of scratch per lane this path used to need).
                accumulate(0, fe_mul2_sum<kChain>(a0, b0, c0, d0));
                if constexpr (!kSkip1) accumulate(1, fe_mul2_sum<kChain>(a1, b1, c1, d1));
                accumulate(2, fe_mul2_sum<kChain>(ai, bi, ci, di));
                auto extend = [](Fe &q0, Fe &q1, const Fe &qi) { // (q0, q1) <- (q(-1), q(2))
                    const Fe t = fe_add(qi, qi);
                    const Fe m1 = fe_carry_pass(fe_sub(fe_add(t, fe_add(q0, q0)), q1));
                    const Fe p2 = fe_carry_pass(fe_sub(fe_add(t, fe_add(q1, q1)), q0));
                    q0 = m1;
                    q1 = p2;
                };
                extend(a0, a1, ai);
                extend(b0, b1, bi);
                extend(c0, c1, ci);
                extend(d0, d1, di);
                fe_pin3(a0, b0, c0);
                fe_pin3(a1, b1, c1);
                accumulate(3, fe_mul2_sum<kChain>(a0, b0, c0, d0));
                accumulate(4, fe_mul2_sum<kChain>(a1, b1, c1, d1));
                continue;
            } else if constexpr (M == 2) {
                Fe l0, h0, l1, h1, m0, k0, m1, k1;
                LoadFactor<0, kR1, kChain>::run(S, b, r, l0, h0);
                LoadFactor<1, kR1, kChain>::run(S, b, r, l1, h1);
                LoadFactor<0, kR1, kChain>::run(S, b2, r, m0, k0);
                LoadFactor<1, kR1, kChain>::run(S, b2, r, m1, k1);
                P[0] = fe_mul2_sum<kChain>(l0, l1, m0, m1);
                if constexpr (!kSkip1) P[1] = fe_mul2_sum<kChain>(h0, h1, k0, k1);
                P[2] = fe_mul2_sum<kChain>(fe_sub(h0, l0), fe_sub(h1, l1), fe_sub(k0, m0), fe_sub(k1, m1));
            } else {
                Fe q0, q1, qi, l2, h2;
                {
                    Fe l0, h0, l1, h1;
                    LoadFactor<0, kR1, kChain>::run(S, b, r, l0, h0);
                    LoadFactor<1, kR1, kChain>::run(S, b, r, l1, h1);
                    q0 = fe_mul<kChain>(l0, l1);
                    q1 = fe_mul<kChain>(h0, h1);
                    qi = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
                }
                fe_pin3(q0, q1, qi);
                LoadFactor<2, kR1, kChain>::run(S, b, r, l2, h2);
                Fe s0, s1, si, m2, k2;
                {
                    Fe l0, h0, l1, h1;
                    LoadFactor<0, kR1, kChain>::run(S, b2, r, l0, h0);
                    LoadFactor<1, kR1, kChain>::run(S, b2, r, l1, h1);
                    s0 = fe_mul<kChain>(l0, l1);
                    s1 = fe_mul<kChain>(h0, h1);
                    si = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
                }
                fe_pin3(s0, s1, si);
                LoadFactor<2, kR1, kChain>::run(S, b2, r, m2, k2);
                P[0] = fe_mul2_sum<kChain>(l2, q0, m2, s0);
                if constexpr (!kSkip1) P[1] = fe_mul2_sum<kChain>(h2, q1, k2, s1);
                P[2] = fe_mul2_sum<kChain>(fe_sub(h2, l2), qi, fe_sub(k2, m2), si);
                // node -1: f2(-1) = 2 lo - hi (re-tightened: the shared reduction needs both operands within 2^29), q(-1) = 2 q(0) + 2 q(inf) - q(1)
                const Fe qm1 = fe_carry_pass(fe_sub(fe_add(fe_add(qi, qi), fe_add(q0, q0)), q1));
                const Fe sm1 = fe_carry_pass(fe_sub(fe_add(fe_add(si, si), fe_add(s0, s0)), s1));
                P[3] = fe_mul2_sum<kChain>(fe_carry_pass(fe_sub(fe_add(l2, l2), h2)), qm1, fe_carry_pass(fe_sub(fe_add(m2, m2), k2)), sm1);
            }
#pragma unroll
            for (int t = 0; t <= M; ++t) {
                if (kSkip1 && t == 1) continue;
                Fe a;
#pragma unroll
                for (int l = 0; l < 9; ++l) a.l[l] = my[(9 * t + l) * kBlock];
                a = fe_add(a, P[t]);
                if (iter & 1u) a = fe_carry_pass(a);
                if ((iter & 31u) == 31u) a = fe_from_fr(fe_to_fr(a));
#pragma unroll
                for (int l = 0; l < 9; ++l) my[(9 * t + l) * kBlock] = a.l[l];
            }
        }
    }
    for (; b < n_pairs; b += stride, ++iter) {
        // Factors are loaded (and bound) in the order the tree consumes them, so at most two lines are live next to the
        // half-products: a0/a1/ai of factors 0,1 are formed before factors 2,3 are touched.
        Fe P[M + 1];
        if constexpr (M == 1) {
            LoadFactor<0, kR1, kChain>::run(S, b, r, P[0], P[1]);
        } else if constexpr (M == 2) {
            Fe l0, h0, l1, h1;
            LoadFactor<0, kR1, kChain>::run(S, b, r, l0, h0);
            LoadFactor<1, kR1, kChain>::run(S, b, r, l1, h1);
            P[0] = fe_mul<kChain>(l0, l1);
            if constexpr (!kSkip1) P[1] = fe_mul<kChain>(h0, h1);
            P[2] = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
        } else if constexpr (M == 3) {
            Fe q0, q1, qi;
            {
                Fe l0, h0, l1, h1;
                LoadFactor<0, kR1, kChain>::run(S, b, r, l0, h0);
                LoadFactor<1, kR1, kChain>::run(S, b, r, l1, h1);
                q0 = fe_mul<kChain>(l0, l1);
                q1 = fe_mul<kChain>(h0, h1);
                qi = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
            }
            fe_pin3(q0, q1, qi);
            const Fe qm1 = fe_carry_pass(fe_sub(fe_add(fe_add(qi, qi), fe_add(q0, q0)), q1)); // q(-1) = 2 q(0) + 2 q(inf) - q(1)
            Fe l2, h2;
            LoadFactor<2, kR1, kChain>::run(S, b, r, l2, h2);
            P[0] = fe_mul<kChain>(l2, q0);
            if constexpr (!kSkip1) P[1] = fe_mul<kChain>(h2, q1);
            P[2] = fe_mul<kChain>(fe_sub(h2, l2), qi);
            P[3] = fe_mul<kChain>(fe_sub(fe_add(l2, l2), h2), qm1); // f2(-1) = 2 lo - hi
        } else {
            static_assert(M == 4, "the tree kernels take products of at most four multiplicands");
            Fe a0, a1, ai, b0, b1, bi;
            {
                Fe l0, h0, l1, h1;
                LoadFactor<0, kR1, kChain>::run(S, b, r, l0, h0);
                LoadFactor<1, kR1, kChain>::run(S, b, r, l1, h1);
                a0 = fe_mul<kChain>(l0, l1);
                a1 = fe_mul<kChain>(h0, h1);
                ai = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
            }
            fe_pin3(a0, a1, ai);
            {
                Fe l2, h2, l3, h3;
                LoadFactor<2, kR1, kChain>::run(S, b, r, l2, h2);
                LoadFactor<3, kR1, kChain>::run(S, b, r, l3, h3);
                b0 = fe_mul<kChain>(l2, l3);
                b1 = fe_mul<kChain>(h2, h3);
                bi = fe_mul<kChain>(fe_sub(h2, l2), fe_sub(h3, l3));
            }
            // a quadratic from its values at 0, 1 and its leading coefficient: q(-1) = 2 q(0) + 2 q(inf) - q(1), q(2) = 2 q(1) + 2 q(inf) - q(0)
            const Fe a2i = fe_add(ai, ai), b2i = fe_add(bi, bi);
            P[0] = fe_mul<kChain>(a0, b0);
            if constexpr (!kSkip1) P[1] = fe_mul<kChain>(a1, b1);
            P[2] = fe_mul<kChain>(ai, bi);
            P[3] = fe_mul<kChain>(fe_carry_pass(fe_sub(fe_add(a2i, fe_add(a0, a0)), a1)), fe_carry_pass(fe_sub(fe_add(b2i, fe_add(b0, b0)), b1)));
            P[4] = fe_mul<kChain>(fe_carry_pass(fe_sub(fe_add(a2i, fe_add(a1, a1)), a0)), fe_carry_pass(fe_sub(fe_add(b2i, fe_add(b1, b1)), b0)));
        }
#pragma unroll
        for (int t = 0; t <= M; ++t) {
            if (kSkip1 && t == 1) continue;
            Fe a;
#pragma unroll
            for (int l = 0; l < 9; ++l) a.l[l] = my[(9 * t + l) * kBlock];
            a = fe_add(a, P[t]);
            // limbs: tightened + two products' limbs < 3 * 2^29 < 2^31, so a carry pass every other iteration suffices
            if (iter & 1u) a = fe_carry_pass(a);
            if ((iter & 31u) == 31u) a = fe_from_fr(fe_to_fr(a)); // keep the top limb far from 2^31 on very long grid-stride loops
#pragma unroll
            for (int l = 0; l < 9; ++l) my[(9 * t + l) * kBlock] = a.l[l];
        }
    }
    // Block sums of the M + 1 nodes TOGETHER: the canonical conversions and the six shuffle steps of the nodes are independent chains
    // that the scheduler interleaves, the wavefronts' sums cross through LDS behind ONE pair of barriers (the running sums' LDS is free
    // by then), and threads 0..M each finish one node.  One node after the other, this epilogue was the last microseconds of every block of
    // a four-multiplicand row -- invisible in the long rounds, visible in the short ones.
#ifndef SC_SERIAL_EPILOGUE // (A/B build)
    (void)sm;
    Fr s[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) {
        if (kSkip1 && t == 1) { s[t] = fr_zero(); continue; }
        Fe a;
#pragma unroll
        for (int l = 0; l < 9; ++l) a.l[l] = my[(9 * t + l) * kBlock];
        s[t] = fe_to_fr(a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int t = 0; t <= M; ++t)
            if (!(kSkip1 && t == 1)) s[t] = fr_add(s[t], fr_shfl_down(s[t], off));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *x = reinterpret_cast<uint32_t *>(lacc); // [wave][node][8]
    __syncthreads();                                   // every thread has read its running sums
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t <= M; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[(wave * (M + 1) + t) * 8 + i] = s[t].v[i];
    }
    __syncthreads();
    if (threadIdx.x <= (uint32_t)M && !(kSkip1 && threadIdx.x == 1)) { // (node 1's row of the partials is left alone: nothing reads it)
        const int t = threadIdx.x;
        Fr acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = x[t * 8 + i];
        for (int w = 1; w < kBlock / 64; ++w) {
            Fr o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = x[(w * (M + 1) + t) * 8 + i];
            acc = fr_add(acc, o);
        }
        fr_store(row + 2 * ((uint64_t)t * pstride), acc);
    }
    __syncthreads(); // (a block that walks several products -- the experiments build's k_round_tree -- reuses the LDS at once)
#else
#pragma unroll
    for (int t = 0; t <= M; ++t) {
        if (kSkip1 && t == 1) continue;
        Fe a;
#pragma unroll
        for (int l = 0; l < 9; ++l) a.l[l] = my[(9 * t + l) * kBlock];
        const Fr s = block_sum(fe_to_fr(a), sm);
        if (threadIdx.x == 0) fr_store(row + 2 * ((uint64_t)t * pstride), s);
    }
#endif
}

template <int M>
__global__ __launch_bounds__(kBlock) void k_prod_tree(const ProdArgs A, const BindConst r, const uint64_t n_pairs,
                                                      uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    __shared__ int32_t rt[kBindLds];
    __shared__ int32_t lacc[9 * (M + 1) * kBlock];
    bind_consts_to_lds(r, rt);
    tree_pass<M>(A.slot, rt, n_pairs, partials + 2 * (uint64_t)blockIdx.x, sm, lacc);
}

#ifdef SC_EXPERIMENTS // cross-check variant and measured negative result: LDS-tiled kernel (SC_KERNEL=2)
// ------------------------------------------------------------------------------------------------
// K4, tiled: the fused bind + product-sum with fine-grained work items staged through LDS.
//
// k_prod_round / k_prod_round_fe give one lane a whole pair: 2 binds and M-1 products for each of M+1 nodes in one
// dependent chain behind ~200 live VGPRs, so only 2-3 waves fit a SIMD and the VALU idles ~45 % of the time
// (profiles: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = 0.40-0.50).  Here a block of 64*(M+1) threads walks tiles of 64
// pairs in two phases:
//   A  one lane per (table slot, entry): load (bind mode: two adjacent entries of the previous table, 64 contiguous
//      bytes per lane, bind r, store the bound entry to HBM), convert to 29-bit limbs and park it in LDS
//      (limb-planar: lane i writes dword i of every limb row -> conflict-free);
//   B  one wavefront per evaluation node, one lane per pair: read (lo, hi) of every slot from LDS with ds_read_b64,
//      form the node's operand (0: lo, 1: hi, inf: hi-lo, -1: 2lo-hi, 2: 2hi-lo, ... all lazy adds) and multiply the
//      M factors; accumulate the lane's running sum for that node across tiles.
// Each lane holds one accumulator and one operand set (~80 VGPRs -> 5-6 waves per SIMD) and the longest dependent
// chain is M-1 products.
// ------------------------------------------------------------------------------------------------
constexpr int kTilePairs = 64;

template <int M>
__global__ __launch_bounds__(64 * (M + 1)) void k_round_tile(const ProdArgs A, const FrHost r32_h, const uint64_t n_pairs,
                                                             uint4 *__restrict__ partials) {
    constexpr int kThreads = 64 * (M + 1);
    constexpr int kEnt = 2 * kTilePairs; // entries per slot per tile
    __shared__ int32_t lds[M * 9 * kEnt]; // [slot][limb][entry]
    const FeU r = feu_from_host(r32_h);
    const int tid = threadIdx.x;
    const int node_idx = tid >> 6; // one wavefront per node
    const int lane = tid & 63;
    const int32_t nv = node_value(node_idx);
    const int n_slots = A.n_slots;
    Fe acc = fe_zero();
    const uint64_t n_tiles = (n_pairs + kTilePairs - 1) / kTilePairs;
    uint32_t iter = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++iter) {
        const uint64_t b0 = tile * kTilePairs;            // first pair of the tile
        const uint64_t ent_left = 2 * (n_pairs - b0);     // valid entries in this tile (>= 2)
        // ---- phase A: stage every slot's 128 entries -------------------------------------------------------
        for (int item = tid; item < n_slots * kEnt; item += kThreads) {
            const int s = item / kEnt, i = item % kEnt; // entry i of the tile = table entry 2*b0 + i
            Fe v = fe_zero();
            if ((uint64_t)i < ent_left) {
                if (A.slot[s].mode == 0) {
                    v = fe_from_fr(fr_load(A.slot[s].src + 2 * (2 * b0 + i)));
                } else {
                    const uint4 *p = A.slot[s].src + 4 * (2 * b0 + i); // previous-table entries 2e, 2e+1: 64 contiguous bytes
                    const Fe e0 = fe_from_fr(fr_load(p)), e1 = fe_from_fr(fr_load(p + 2));
                    const Fr c = fe_to_fr(fe_add(e0, fe_mul_u(fe_sub(e1, e0), r)));
                    fr_store(A.slot[s].dst + 2 * (2 * b0 + i), c); // tables stay canonical in the reference layout
                    v = fe_from_fr(c);
                }
            }
            int32_t *row = lds + (s * 9) * kEnt + i;
#pragma unroll
            for (int l = 0; l < 9; ++l) row[l * kEnt] = v.l[l];
        }
        __syncthreads();
        // ---- phase B: this wavefront's node, this lane's pair ------------------------------------------------
        {
            Fe prod;
            bool first = true;
            for (int s = 0; s < n_slots; ++s) {
                Fe lo, hi;
                const int32_t *row = lds + (s * 9) * kEnt + 2 * lane;
#pragma unroll
                for (int l = 0; l < 9; ++l) {
                    const int2 w = *reinterpret_cast<const int2 *>(row + l * kEnt);
                    lo.l[l] = w.x;
                    hi.l[l] = w.y;
                }
                Fe val;
                if (nv == 0) val = lo;
                else if (nv == 1) val = hi;
                else if (nv == kNodeInf) val = fe_sub(hi, lo);
                else if (nv == -1) val = fe_sub(fe_add(lo, lo), hi);
                else if (nv == 2) val = fe_sub(fe_add(hi, hi), lo);
                else { // further out: walk along the line, re-tightening the limbs before every step
                    const Fe step = fe_sub(hi, lo);
                    if (nv > 0) {
                        val = fe_sub(fe_add(hi, hi), lo);
                        for (int32_t c = 2; c < nv; ++c) val = fe_add(fe_carry_pass(val), step);
                    } else {
                        val = fe_sub(fe_add(lo, lo), hi);
                        for (int32_t c = -1; c > nv; --c) val = fe_sub(fe_carry_pass(val), step);
                    }
                }
                uint32_t k = 0;
                if (first) { prod = (nv == 0 || nv == 1) ? val : fe_carry_pass(val); k = 1; first = false; }
                for (; k < A.slot[s].exp; ++k) prod = fe_mul(val, prod);
            }
            if (b0 + lane < n_pairs) acc = fe_carry_pass(fe_add(acc, prod));
            if ((iter & 31u) == 31u) acc = fe_from_fr(fe_to_fr(acc));
        }
        __syncthreads(); // LDS is overwritten by the next tile
    }
    // wavefront reduction of this node's sums, one partial per block
    Fr sum = fe_to_fr(acc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum = fr_add(sum, fr_shfl_down(sum, off));
    if (lane == 0) fr_store(partials + 2 * ((uint64_t)node_idx * gridDim.x + blockIdx.x), sum);
}

#endif // SC_EXPERIMENTS

#ifdef SC_EXPERIMENTS // the previous form of the big-round kernels (SC_SPLIT=0) and the in-kernel finalize experiment built on it
// every product of the round in one launch (RoundArgs in kernels.h).
// Experiments build only -- R.fin.enabled: the round's finalize step inside the launch.  A separate k_finalize launch costs a dispatch gap (~5 us) plus ~20 us for one block to add up 768 x 14
// partials.  Here the blocks that finish last do it in two levels: the last block of every group of kFinGroup blocks (by block index;
// an arrival counter per group) adds the group's partials into one set, and the block that completes the last group combines the
// ~24 group sets into the message (finalize_body), publishes it and resets the counters for the next launch.  Everything the
// combining block reads was released (agent scope) by its writer before the counter it acquired was incremented.
constexpr int kFinGroup = 32;
__global__ __launch_bounds__(kBlock, 3) void k_round_tree(const RoundArgs R, const BindConst r, const uint64_t n_pairs,
                                                       uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    __shared__ int32_t rt[kBindLds];
    __shared__ int32_t lacc[9 * 5 * kBlock];
#ifdef SC_EXPERIMENTS
    __shared__ uint32_t role_sh;
#endif
    bind_consts_to_lds(r, rt);
    const int n = R.n_prod;
    // start product rotated by dispatch slot (blockIdx & 7 = XCD): multiplier-bound and HBM-bound products overlap across an XCD's CUs
    int k = (int)((blockIdx.x & 7u) % (uint32_t)n);
    for (int i = 0; i < n; ++i) {
        const TreeProd &T = R.prod[k];
        uint4 *row = partials + 2 * (T.partial_off + (uint64_t)blockIdx.x);
        switch (T.M) {
        case 1: tree_pass<1>(T.slot, rt, n_pairs, row, sm, lacc); break;
        case 2: tree_pass<2>(T.slot, rt, n_pairs, row, sm, lacc); break;
        case 3: tree_pass<3>(T.slot, rt, n_pairs, row, sm, lacc); break;
        default: tree_pass<4>(T.slot, rt, n_pairs, row, sm, lacc); break;
        }
        if (++k == n) k = 0;
    }
#ifdef SC_EXPERIMENTS // measured negative result (SC_FUSED_FIN=1): +45 us per launch -- every block's agent-scope release is a write-back of the
                      // XCD's L2, full of freshly bound table lines in rounds >= 2 -- against the ~25 us of a separate k_finalize launch
    if (!R.fin.enabled) return;
    // ---- level 1: the last block of this group adds up the group's partials ---------------------------------------------------
    const uint32_t G = gridDim.x, n_groups = (G + kFinGroup - 1) / kFinGroup;
    const uint32_t g = blockIdx.x / kFinGroup, g_first = g * kFinGroup, g_size = min((uint32_t)kFinGroup, G - g_first);
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t d = __hip_atomic_fetch_add(R.fin.counters + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role_sh = d == g_size - 1 ? 1u : 0u;
        if (role_sh) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!role_sh) return;
    {
        // items (product, node, quarter q of the group): one lane adds up to 8 partials, four adjacent lanes combine
        int n_items = 0;
        for (int q = 0; q < n; ++q) n_items += (int)R.prod[q].M + 1;
        for (int it0 = 0; it0 < 4 * n_items; it0 += kBlock) { // block-uniform trip count
            const int it = it0 + (int)threadIdx.x;
            const bool live = it < 4 * n_items;
            int c = live ? it >> 2 : 0, q4 = it & 3, pk = 0;
            while (c > (int)R.prod[pk].M) { c -= (int)R.prod[pk].M + 1; ++pk; } // c = node of product pk
            const uint4 *base = partials + 2 * (R.prod[pk].partial_off + (uint64_t)c * G + g_first);
            Fr acc = fr_zero();
            if (live)
                for (uint32_t b = (uint32_t)q4 * 8; b < min((uint32_t)q4 * 8 + 8, g_size); ++b) acc = fr_add(acc, fr_load(base + 2 * b));
            acc = fr_add(acc, fr_shfl_down(acc, 2));
            acc = fr_add(acc, fr_shfl_down(acc, 1));
            if (live && q4 == 0) fr_store(R.fin.partials2 + 2 * (R.prod[pk].partial_off + (uint64_t)c * n_groups + g), acc);
        }
    }
    // ---- level 2: the block that completes the last group writes the message --------------------------------------------------
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t d = __hip_atomic_fetch_add(R.fin.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role_sh = d == n_groups - 1 ? 2u : 0u;
        if (role_sh) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (role_sh != 2u) return;
    for (uint32_t i = threadIdx.x; i <= n_groups; i += kBlock) // the counters start the next launch at zero
        __hip_atomic_store(R.fin.counters + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto prod_of = [&](int q) -> FinProd {
        FinProd f;
        f.M = R.prod[q].M;
        f.pad = 0;
        f.partial_off = R.prod[q].partial_off;
        f.w_off = R.fin.w_off[q];
        return f;
    };
    finalize_body<kBlock>(prod_of, R.fin.Wm, n, R.fin.D, (int)n_groups, R.fin.partials2, reinterpret_cast<uint4 *>(lacc), R.fin.out, R.fin.out_wide,
                          R.fin.h_out, R.fin.h_flag, R.fin.seq, 1);
#endif // SC_EXPERIMENTS
}
#endif // SC_EXPERIMENTS

// The same round with one product per block row (grid.y = product): for the big rounds with few pairs per lane the time of a launch is
// the longest dependent chain of one lane -- all products of a pair, 41 Montgomery products for config 3 -- and splitting by product
// cuts it to the longest product's (19) at the price of more, smaller blocks.  Same slots, same partial layout (gridDim.x blocks per
// product), so the finalize step does not change.
__global__ __launch_bounds__(kBlock, 3) void k_round1_tree_split(const RoundArgs R, const uint64_t n_pairs, uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    __shared__ int32_t rt[kBindLds]; // (unused: the factor loader's signature)
    __shared__ int32_t lacc[9 * 5 * kBlock];
    const TreeProd &T = R.prod[blockIdx.y];
    uint4 *row = partials + 2 * (T.partial_off + (uint64_t)R.part_block0 + (uint64_t)blockIdx.x);
#ifdef SC_NO_CHAIN_R1 // A/B build
    constexpr bool kC1 = false;
#else
    constexpr bool kC1 = true;
#endif
    switch (T.M) {
    case 1: tree_pass<1, true, kC1>(T.slot, rt, n_pairs, row, sm, lacc, R.part_stride); break;
    case 2: tree_pass<2, true, kC1>(T.slot, rt, n_pairs, row, sm, lacc, R.part_stride); break;
    case 3: tree_pass<3, true, kC1>(T.slot, rt, n_pairs, row, sm, lacc, R.part_stride); break;
    default: tree_pass<4, true, kC1>(T.slot, rt, n_pairs, row, sm, lacc, R.part_stride); break;
    }
}
// kChain: single-chain multiply-adds (fe_device.hpp) -- the instantiation for a proof's first binding round, whose sources are canonical.
// (Round 4 measured instantiations for launches whose products all have at most three multiplicands -- four nodes of running sums, one
// pair per iteration, 113-116 registers, FOUR resident blocks per CU -- on config 4's shapes: no change beyond box noise,
// profiles/r4c_config4_m3_occupancy_ab.txt; those rounds already move their real bytes at 5.2-5.3 TB/s.  Not kept.)
template <bool kChain, bool kSkip1>
__global__ __launch_bounds__(kBlock, 3) void k_round_tree_split(const RoundArgs R, const BindConst r, const uint64_t n_pairs, uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    __shared__ int32_t rt[kBindLds];
    __shared__ int32_t lacc[9 * 5 * kBlock];
    bind_consts_to_lds(r, rt);
    const TreeProd &T = R.prod[blockIdx.y];
    uint4 *row = partials + 2 * (T.partial_off + (uint64_t)blockIdx.x);
    switch (T.M) {
    case 1: tree_pass<1, false, kChain, kSkip1>(T.slot, rt, n_pairs, row, sm, lacc); break;
    case 2: tree_pass<2, false, kChain, kSkip1>(T.slot, rt, n_pairs, row, sm, lacc); break;
    case 3: tree_pass<3, false, kChain, kSkip1>(T.slot, rt, n_pairs, row, sm, lacc); break;
    default: tree_pass<4, false, kChain, kSkip1>(T.slot, rt, n_pairs, row, sm, lacc); break;
    }
}

#ifdef SC_EXPERIMENTS
// Round 1 of a proof as its own instantiation: no bind, canonical inputs only.  Without the bind path's registers the kernel affords
// two pairs per iteration for products of FOUR multiplicands too (five more shared reductions per two pairs).
__global__ __launch_bounds__(kBlock, 3) void k_round1_tree(const RoundArgs R, const uint64_t n_pairs, uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    __shared__ int32_t rt[kBindLds]; // (unused: the factor loader's signature)
    __shared__ int32_t lacc[9 * 5 * kBlock];
    const int n = R.n_prod;
    int k = (int)((blockIdx.x & 7u) % (uint32_t)n);
    for (int i = 0; i < n; ++i) {
        const TreeProd &T = R.prod[k];
        uint4 *row = partials + 2 * (T.partial_off + (uint64_t)blockIdx.x);
        switch (T.M) {
        case 1: tree_pass<1, true>(T.slot, rt, n_pairs, row, sm, lacc); break;
        case 2: tree_pass<2, true>(T.slot, rt, n_pairs, row, sm, lacc); break;
        case 3: tree_pass<3, true>(T.slot, rt, n_pairs, row, sm, lacc); break;
        default: tree_pass<4, true>(T.slot, rt, n_pairs, row, sm, lacc); break;
        }
        if (++k == n) k = 0;
    }
}
#endif // SC_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#ifdef SC_EXPERIMENTS
template <int M>
static hipError_t launch_prod_round_t(const ProdArgs &args, const FrHost &r, uint64_t n_pairs, FrHost *d_partials, int grid,
                                      hipStream_t stream) {
    hipLaunchKernelGGL(k_prod_round<M>, dim3(grid), dim3(kBlock), 0, stream, args, r, n_pairs, (uint4 *)d_partials);
    return hipGetLastError();
}

#endif

template <int M>
static hipError_t launch_prod_round_fe_t(const ProdArgs &args, const FrHost &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                                         hipStream_t stream) {
    hipLaunchKernelGGL(k_prod_round_fe<M>, dim3(grid), dim3(kBlock), 0, stream, args, r32, n_pairs, (uint4 *)d_partials);
    return hipGetLastError();
}

hipError_t launch_prod_round_fe(int M, const ProdArgs &args, const FrHost &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                                hipStream_t stream) {
    switch (M) {
    case 1: return launch_prod_round_fe_t<1>(args, r32, n_pairs, d_partials, grid, stream);
    case 2: return launch_prod_round_fe_t<2>(args, r32, n_pairs, d_partials, grid, stream);
    case 3: return launch_prod_round_fe_t<3>(args, r32, n_pairs, d_partials, grid, stream);
    case 4: return launch_prod_round_fe_t<4>(args, r32, n_pairs, d_partials, grid, stream);
    case 5: return launch_prod_round_fe_t<5>(args, r32, n_pairs, d_partials, grid, stream);
    case 6: return launch_prod_round_fe_t<6>(args, r32, n_pairs, d_partials, grid, stream);
    case 7: return launch_prod_round_fe_t<7>(args, r32, n_pairs, d_partials, grid, stream);
    case 8: return launch_prod_round_fe_t<8>(args, r32, n_pairs, d_partials, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int M>
static hipError_t launch_prod_tree_t(const ProdArgs &args, const BindConst &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(k_prod_tree<M>, dim3(grid), dim3(kBlock), 0, stream, args, r32, n_pairs, (uint4 *)d_partials);
    return hipGetLastError();
}

hipError_t launch_prod_tree(int M, const ProdArgs &args, const BindConst &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                            hipStream_t stream) {
    switch (M) {
    case 1: return launch_prod_tree_t<1>(args, r32, n_pairs, d_partials, grid, stream);
    case 2: return launch_prod_tree_t<2>(args, r32, n_pairs, d_partials, grid, stream);
    case 3: return launch_prod_tree_t<3>(args, r32, n_pairs, d_partials, grid, stream);
    case 4: return launch_prod_tree_t<4>(args, r32, n_pairs, d_partials, grid, stream);
    default: return launch_prod_tree_wide(M, args, r32, n_pairs, d_partials, grid, stream); // 5..8 factors: kernels_wide.hip
    }
}

hipError_t launch_round_tree(const RoundArgs &args, const BindConst &r32, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream, bool split,
                             bool skip1) {
    size_t extra_lds = 0;
#ifdef SC_EXPERIMENTS // SC_EXTRA_LDS: unused dynamic LDS per block, to lower the number of resident blocks per CU (occupancy experiments)
    static const size_t env_lds = [] {
        const char *e = std::getenv("SC_EXTRA_LDS");
        return e ? (size_t)std::strtoul(e, nullptr, 10) : (size_t)0;
    }();
    extra_lds = env_lds;
#endif
    bool round1 = args.fin.enabled == 0; // every factor read in place from a canonical table: the round-1 instantiation
    for (int q = 0; q < args.n_prod && round1; ++q)
        for (uint32_t f = 0; f < args.prod[q].M; ++f) round1 = round1 && args.prod[q].slot[f].mode == 0 && args.prod[q].slot[f].src_top == nullptr;
#ifdef SC_EXPERIMENTS
    if (!split) {
        if (skip1) return hipErrorInvalidValue;
        if (round1 && extra_lds == 0) hipLaunchKernelGGL(k_round1_tree, dim3(grid), dim3(kBlock), 0, stream, args, n_pairs, (uint4 *)d_partials);
        else hipLaunchKernelGGL(k_round_tree, dim3(grid), dim3(kBlock), extra_lds, stream, args, r32, n_pairs, (uint4 *)d_partials);
        return hipGetLastError();
    }
#else
    (void)split;
    (void)extra_lds;
#endif
    // the first binding round reads canonical tables like round 1 does (every source without a limb-8 array): the single-chain instantiation
#ifdef SC_NO_CHAIN_R2 // A/B build: the single-chain instantiation for round 1 only
    bool canonical_sources = false;
#else
    bool canonical_sources = !round1;
#endif
    for (int q = 0; q < args.n_prod && canonical_sources; ++q)
        for (uint32_t f = 0; f < args.prod[q].M; ++f) canonical_sources = canonical_sources && args.prod[q].slot[f].src_top == nullptr;
    const dim3 g(grid, args.n_prod), b(kBlock);
    uint4 *const part = (uint4 *)d_partials;
    plan_hit(round1 ? kPlanBigMergedRound1 : canonical_sources ? kPlanBigMergedBindChain : kPlanBigMergedBind);
    if (skip1) plan_hit(kPlanBigClaimIdentity);
    if (round1) {
        if (skip1) return hipErrorInvalidValue; // (round 1 has no previous round)
        hipLaunchKernelGGL(k_round1_tree_split, g, b, 0, stream, args, n_pairs, part);
    } else if (canonical_sources) {
        if (skip1) hipLaunchKernelGGL((k_round_tree_split<true, true>), g, b, 0, stream, args, r32, n_pairs, part);
        else hipLaunchKernelGGL((k_round_tree_split<true, false>), g, b, 0, stream, args, r32, n_pairs, part);
    } else {
        if (skip1) hipLaunchKernelGGL((k_round_tree_split<kChainDefault, true>), g, b, 0, stream, args, r32, n_pairs, part);
        else hipLaunchKernelGGL((k_round_tree_split<kChainDefault, false>), g, b, 0, stream, args, r32, n_pairs, part);
    }
    return hipGetLastError();
}

#ifdef SC_EXPERIMENTS
template <int M>
static hipError_t launch_round_tile_t(const ProdArgs &args, const FrHost &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                                      hipStream_t stream) {
    hipLaunchKernelGGL(k_round_tile<M>, dim3(grid), dim3(64 * (M + 1)), 0, stream, args, r32, n_pairs, (uint4 *)d_partials);
    return hipGetLastError();
}

int grid_for_tiles(uint64_t n_pairs) {
    uint64_t g = (n_pairs + kTilePairs - 1) / kTilePairs;
    if (g < 1) g = 1;
    if (g > (uint64_t)kMaxGrid) g = kMaxGrid;
    return (int)g;
}

hipError_t launch_round_tile(int M, const ProdArgs &args, const FrHost &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                             hipStream_t stream) {
    switch (M) {
    case 1: return launch_round_tile_t<1>(args, r32, n_pairs, d_partials, grid, stream);
    case 2: return launch_round_tile_t<2>(args, r32, n_pairs, d_partials, grid, stream);
    case 3: return launch_round_tile_t<3>(args, r32, n_pairs, d_partials, grid, stream);
    case 4: return launch_round_tile_t<4>(args, r32, n_pairs, d_partials, grid, stream);
    case 5: return launch_round_tile_t<5>(args, r32, n_pairs, d_partials, grid, stream);
    case 6: return launch_round_tile_t<6>(args, r32, n_pairs, d_partials, grid, stream);
    case 7: return launch_round_tile_t<7>(args, r32, n_pairs, d_partials, grid, stream);
    case 8: return launch_round_tile_t<8>(args, r32, n_pairs, d_partials, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_prod_round(int M, const ProdArgs &args, const FrHost &r, uint64_t n_pairs, FrHost *d_partials, int grid,
                             hipStream_t stream) {
    switch (M) {
    case 1: return launch_prod_round_t<1>(args, r, n_pairs, d_partials, grid, stream);
    case 2: return launch_prod_round_t<2>(args, r, n_pairs, d_partials, grid, stream);
    case 3: return launch_prod_round_t<3>(args, r, n_pairs, d_partials, grid, stream);
    case 4: return launch_prod_round_t<4>(args, r, n_pairs, d_partials, grid, stream);
    case 5: return launch_prod_round_t<5>(args, r, n_pairs, d_partials, grid, stream);
    case 6: return launch_prod_round_t<6>(args, r, n_pairs, d_partials, grid, stream);
    case 7: return launch_prod_round_t<7>(args, r, n_pairs, d_partials, grid, stream);
    case 8: return launch_prod_round_t<8>(args, r, n_pairs, d_partials, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

#endif // SC_EXPERIMENTS

} // namespace scd
