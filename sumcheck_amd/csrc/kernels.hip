// kernels.hip -- gfx950 kernels of the sumcheck prover hot path.
//
// K2/K3/K4 of SURVEY.md 2.2: the per-round sum over the hypercube (reference
// src/ml_sumcheck/protocol/prover.rs:110-148) fused with the previous challenge's bind
// (prover.rs:84-89 -> DenseMultilinearExtension::fix_variables).  Integer modular arithmetic on the
// VALU; no MFMA (this is not a dense contraction).
//
// Work decomposition (per product k with M multiplicands, per round with n_pairs hypercube points):
//   lane b owns pair b: entries (2b, 2b+1) of this round's tables.  In fused mode it reads the four
//   entries (4b..4b+3) of the previous round's table, binds r twice, stores the two new entries and
//   keeps them in registers for the sum -- each table is read once and written once per round.
//   Only the M+1 evaluations that determine this product's degree-M round polynomial are summed;
//   the coefficient c_k and the extension to the message's deg+1 points are applied once per round
//   in k_finalize (exact field arithmetic => the same canonical bits as the reference's
//   per-point evaluation at all deg+1 points).
//   Per-lane accumulators -> wave64 shuffle reduction -> LDS across the 4 waves -> one partial per
//   block -> k_finalize.
//
// Kernels, by role:
//   k_round_tree_split, k_round1_tree_split   production big-round kernels: ONE launch per round over all products (each <= 4
//                      multiplicands), one product per block row: fe_device.hpp carry-free arithmetic, constant-multiplier bind,
//                      evaluation nodes 0,1,inf,-1,2, static product tree, running sums in LDS, F29 internal table format.
//   k_round_tree, k_round1_tree   the previous form (every block walks all products): experiments build, SC_SPLIT=0.
//   k_prod_tree<M>     the same pass for one product per launch (SC_MERGE=0, or more than kMaxRoundProds products).
//   k_prod_round_fe<M> fe_device.hpp arithmetic node by node (5..8 multiplicands; SC_KERNEL=0 SC_FE=1 as a cross-check).
//   k_prod_round<M>    saturated 8 x u32 Comba arithmetic (SC_KERNEL=0 SC_FE=0) -- kept as a parity cross-check.
//   k_round_tile<M>    LDS-tiled variant (SC_KERNEL=2), a measured negative result.
//   k_fold_multi<L>    evaluation at a point: all tables, L <= 3 variables per pass (sc_poly_evaluate).
//   k_sum_generic/k_fix  any M, any aliasing pattern; used beyond kMaxFusedM and for > 32 tables.
//   k_fix_multi + k_sum_combos   latency-oriented pair for rounds with <= 2^14 pairs (kSmallRoundPairs).
//   k_tail_rounds      every round with <= 2048 pairs in one persistent launch (grid barrier, host mailbox).
//   k_finalize_mb, k_finalize   partial sums -> round message (Lagrange matrix, c_k, sum over products).
#include "kernel_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace scd {
#ifdef SC_FIN_CLOCKS // tools/build_variant.sh fin_clocks -DSC_FIN_CLOCKS: where k_finalize's time goes (100 MHz wall clock)
__device__ uint64_t g_fin_clk[12];
#define FIN_STAMP(i)                                                                                                                      \
    do {                                                                                                                                  \
        if (threadIdx.x == 0) g_fin_clk[i] = wall_clock64();                                                                              \
    } while (0)
#else
#define FIN_STAMP(i)
#endif
} // namespace scd
#include "finalize_device.hpp"
namespace scd {
// ------------------------------------------------------------------------------------------------
// Generic product-sum (any number of multiplicands): grid.y = evaluation point t, tables already bound.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sum_generic(const uint4 *const *__restrict__ cur_tables,
                                                        const uint32_t *__restrict__ slot_table,
                                                        const uint32_t *__restrict__ slot_exp, const int n_slots, const int M,
                                                        const uint64_t n_pairs, uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    const uint32_t t = blockIdx.y;
    const int32_t nv = node_value((int)t);
    const Fr tf = node_constant(nv);
    Fr acc = fr_zero();
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_pairs; b += stride) {
        Fr prod = fr_one();
        for (int s = 0; s < n_slots; ++s) {
            const uint4 *p = cur_tables[slot_table[s]] + 4 * b;
            const Fr lo = fr_load(p), hi = fr_load(p + 2);
            Fr val;
            if (nv == 0) val = lo;
            else if (nv == 1) val = hi;
            else if (nv == kNodeInf) val = fr_sub(hi, lo);
            else val = fr_add(lo, fr_mul(tf, fr_sub(hi, lo)));
            for (uint32_t k = 0; k < slot_exp[s]; ++k) prod = fr_mul(prod, val);
        }
        acc = fr_add(acc, prod);
    }
    const Fr s = block_sum(acc, sm);
    if (threadIdx.x == 0) fr_store(partials + 2 * ((uint64_t)t * gridDim.x + blockIdx.x), s);
}

// ------------------------------------------------------------------------------------------------
// K3: stand-alone bind of one variable: out[b] = in[2b] + r*(in[2b+1]-in[2b])
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_fix(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const FrHost r_h,
                                                const uint64_t n_out) {
    const FrU r = fru_from_host(r_h);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_out; b += stride) {
        const uint4 *p = src + 4 * b;
        const Fr lo = fr_load(p), hi = fr_load(p + 2);
        fr_store(dst + 2 * b, fr_add(lo, fr_mul_u(fr_sub(hi, lo), r)));
    }
}

// Evaluate-at-a-point pass (kernels.h, FoldArgs): lane i folds entries [i << L, (i + 1) << L) of table blockIdx.y over L variables,
// LSB first (ark-poly's fix_variables order), in carry-free arithmetic; one canonical element out.
template <int L>
__global__ __launch_bounds__(kBlock) void k_fold_multi(const FoldArgs A, const uint64_t n_out) {
    const uint4 *__restrict__ src = A.src[blockIdx.y];
    uint4 *__restrict__ dst = A.dst[blockIdx.y];
    FeU r[L];
#pragma unroll
    for (int l = 0; l < L; ++l) r[l] = feu_from_host(A.r32[l]);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_out; i += stride) {
        Fe v[1 << L];
        const uint4 *p = src + 2 * (i << L);
#pragma unroll
        for (int j = 0; j < (1 << L); ++j) v[j] = fe_from_fr(fr_load(p + 2 * j));
#pragma unroll
        for (int l = 0; l < L; ++l) {
#pragma unroll
            for (int j = 0; j < (1 << (L - 1 - l)); ++j)
                v[j] = fe_carry_pass(fe_add(v[2 * j], fe_mul_u(fe_sub(v[2 * j + 1], v[2 * j]), r[l])));
        }
        fr_store(dst + 2 * i, fe_to_fr(v[0]));
    }
}

hipError_t launch_fold_multi(const FoldArgs &args, int levels, int n_tables, uint64_t n_out, hipStream_t stream) {
    plan_hit(kPlanFoldMulti);
    const dim3 grid(grid_for_pairs(n_out), n_tables);
    switch (levels) {
    case 1: hipLaunchKernelGGL(k_fold_multi<1>, grid, dim3(kBlock), 0, stream, args, n_out); break;
    case 2: hipLaunchKernelGGL(k_fold_multi<2>, grid, dim3(kBlock), 0, stream, args, n_out); break;
    case 3: hipLaunchKernelGGL(k_fold_multi<3>, grid, dim3(kBlock), 0, stream, args, n_out); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// out[i] = s * in[i]   (start_phase2_sumcheck's f3 * f2(u), reference src/gkr_round_sumcheck/mod.rs:71-75)
__global__ __launch_bounds__(kBlock) void k_scale(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const FrHost s_h,
                                                  const uint64_t n) {
    const FrU s = fru_from_host(s_h);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        fr_store(dst + 2 * i, fr_mul_u(fr_load(src + 2 * i), s));
}

// ------------------------------------------------------------------------------------------------
// Small rounds (<= kSmallRoundPairs pairs): a lane's dependent chain in k_prod_round (2 binds + M-1 products for
// each of M+1 points, ~25 Montgomery products = ~30 us) is the whole kernel time, so the work is split finer:
// one launch binds every table (one product deep), one launch computes every (product, point) combination with
// one lane per (combination, pair) (M-1 products deep).
// ------------------------------------------------------------------------------------------------
// Pipelined rounds: holds the stream until the host has published challenge number `want` (a system-scope word in host-mapped
// memory), then copies the challenge from the host-mapped mailbox into device memory, so that the round's kernels behind it
// read it like any other device data (thousands of blocks fetching it over PCIe cost 30-100 us per round).  One wavefront,
// lane 0 polls with s_sleep back-off; the spin is bounded (~2^22 polls, seconds) so a host that never answers cannot hang the
// queue -- the kernels behind it then run on a stale challenge, and the give-up marker makes the host discard their result.
__global__ void k_wait_challenge(uint32_t *__restrict__ flag, const uint32_t want, const uint32_t max_spins,
                                 const uint64_t *__restrict__ mail_host, uint64_t *__restrict__ mail_dev) {
    if (threadIdx.x == 0) {
        bool seen = false;
        for (uint32_t spin = 0; spin < max_spins && !seen; ++spin) {
            seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == want;
            if (!seen) __builtin_amdgcn_s_sleep(8);
        }
        // gave up: tell the host (flag[1]) so that it rejects whatever the kernels behind this one publish
        if (!seen) __hip_atomic_store(flag + 1, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (threadIdx.x < 4) mail_dev[threadIdx.x] = __hip_atomic_load(mail_host + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
uint32_t wait_spins_default() { return (uint32_t)policy(kPolWaitSpins); } // (tests shorten the bound to exercise the give-up path)
hipError_t launch_wait_challenge(uint32_t *flag_dev, uint32_t want, const FrHost *mail_host_dev, FrHost *mail_dev, hipStream_t stream,
                                 uint32_t spins_override) {
    const uint32_t max_spins = spins_override ? spins_override : wait_spins_default();
    hipLaunchKernelGGL(k_wait_challenge, dim3(1), dim3(64), 0, stream, flag_dev, want, max_spins, reinterpret_cast<const uint64_t *>(mail_host_dev),
                       reinterpret_cast<uint64_t *>(mail_dev));
    return hipGetLastError();
}

// r_mail != nullptr: the launch was enqueued before the challenge existed (pipelined rounds, protocol.hip); the challenge is read from
// the device-memory mailbox that k_wait_challenge, the kernel in front of this one, filled
__global__ __launch_bounds__(kBlock) void k_fix_multi(const TablePtrs tp, const FrHost r_h, const FrHost *__restrict__ r_mail, const uint64_t n_out) {
    const FrU r = fru_from_host(r_mail ? *r_mail : r_h); // uniform: scalar loads either way
    const FeU r32 = feu_shl5(r.v);                       // carry-free bind: one chain of multiply-adds, no carry instruction (fe_device.hpp)
    const uint4 *__restrict__ src = tp.src[blockIdx.y];
    uint4 *__restrict__ dst = tp.dst[blockIdx.y];
    const int32_t *__restrict__ stop = tp.src_top[blockIdx.y];
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_out; b += stride) {
        const uint4 *p = src + 4 * b;
        Fe lo, hi;
        if (stop) { // first latency-bound round after the big rounds: the table arrives in F29 (used as it is), leaves canonical
            const int2 t = *reinterpret_cast<const int2 *>(stop + 2 * b);
            lo = fe_load_f29(src, 2 * b, t.x);
            hi = fe_load_f29(src, 2 * b + 1, t.y);
        } else {
            lo = fe_from_fr(fr_load(p));
            hi = fe_from_fr(fr_load(p + 2));
        }
        fr_store(dst + 2 * b, fe_to_fr(fe_add(lo, fe_mul_u<true>(fe_sub(hi, lo), r32))));
    }
}

__global__ __launch_bounds__(kBlock) void k_f29_to_sat(const uint4 *__restrict__ src, const int32_t *__restrict__ stop, uint4 *__restrict__ dst,
                                                       const uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        fr_store(dst + 2 * i, fe_to_fr(fe_load_f29(src, i, stop[i])));
}

// The product of one (product, node) combination at pair b: prod_s line_s(node)^exp_s over the combination's slots.
// A latency-bound round is a chain of dependent memory accesses per slot -- which table (slot_table), where it is (tab), its two entries
// -- followed by a dependent multiplication; walking the slots one after the other made a late round's sums cost ~3 us PER SLOT
// (profiles/r3f_tail_clocks.txt: 9.2 us for four multiplicands against 3.3 us for two, with a single pair).  So the slots' entries are
// fetched four at a time BEFORE anything is multiplied: the loads of a group are in flight together.
template <typename SlotsT, typename TabFn>
__device__ __forceinline__ Fr combo_product(const TabFn &tab, const Combo &c, const SlotsT slot_table, const SlotsT slot_exp, const uint64_t b,
                                            const int32_t nv, const Fr &tf) {
    Fr prod = fr_zero();
    bool first = true;
    for (uint32_t s0 = 0; s0 < c.n_slots; s0 += 4) {
        const uint32_t ns = c.n_slots - s0 < 4u ? c.n_slots - s0 : 4u;
        Fr lo[4], hi[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            if (q < ns) {
                const uint4 *p = tab(slot_table[c.slot_off + s0 + q]) + 4 * b;
                if (nv != 1) lo[q] = fr_load(p);
                if (nv != 0) hi[q] = fr_load(p + 2);
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            if (q < ns) {
                Fr val;
                if (nv == 0) val = lo[q];
                else if (nv == 1) val = hi[q];
                else if (nv == kNodeInf) val = fr_sub(hi[q], lo[q]);
                else if (nv == -1) val = fr_sub(fr_add(lo[q], lo[q]), hi[q]); // the line at -1 and at 2: two modular adds, no product
                else if (nv == 2) val = fr_sub(fr_add(hi[q], hi[q]), lo[q]);
                else val = fr_add(lo[q], fr_mul(fr_sub(hi[q], lo[q]), tf));
                uint32_t k = 0;
                if (first) { prod = val; k = 1; first = false; }
                for (const uint32_t e = slot_exp[c.slot_off + s0 + q]; k < e; ++k) prod = fr_mul(prod, val);
            }
        }
    }
    return prod;
}

// combo_product in CARRY-FREE arithmetic (fe_device.hpp), for products of at most kMaxFusedM multiplicands: a dependent chain of
// saturated Comba products costs a lone wavefront ~2 us each (every multiply-add drags a carry instruction with wait states behind it);
// the 9 x 29-bit product is one chain of 153 back-to-back multiply-adds.  The result carries 2^(-5(M-1)) like the big rounds' partial
// sums do (fe_device.hpp: Montgomery radix 2^261); the finalize step's scaled weights remove it.
template <typename SlotsT, typename TabFn>
__device__ __forceinline__ Fe combo_product_fe(const TabFn &tab, const Combo &c, const SlotsT slot_table, const SlotsT slot_exp, const uint64_t b,
                                               const int32_t nv) {
    Fe prod = fe_zero();
    bool first = true;
    for (uint32_t s0 = 0; s0 < c.n_slots; s0 += 4) {
        const uint32_t ns = c.n_slots - s0 < 4u ? c.n_slots - s0 : 4u;
        Fr lo[4], hi[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            if (q < ns) {
                const uint4 *p = tab(slot_table[c.slot_off + s0 + q]) + 4 * b;
                if (nv != 1) lo[q] = fr_load(p);
                if (nv != 0) hi[q] = fr_load(p + 2);
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            if (q < ns) {
                Fe val;
                if (nv == 0) val = fe_from_fr(lo[q]);
                else if (nv == 1) val = fe_from_fr(hi[q]);
                else val = fe_line(fe_from_fr(lo[q]), fe_from_fr(hi[q]), nv);
                uint32_t k = 0;
                if (first) { prod = val; k = 1; first = false; }
                for (const uint32_t e = slot_exp[c.slot_off + s0 + q]; k < e; ++k) prod = fe_mul<true>(val, prod);
            }
        }
    }
    return prod;
}

// one (product, node) combination over the block's pairs; the metadata comes from device memory or from a kernel argument.
// `tab(u)` gives table u's current evaluations; (vbx, vgx) = this block's index and the block count along the pair axis (the
// launch's blockIdx.x / gridDim.x, or the virtual ones of the persistent tail kernel).
template <typename SlotsT, typename TabFn>
__device__ __forceinline__ void sum_combo_body(const TabFn &tab, const Combo c, const SlotsT slot_table, const SlotsT slot_exp,
                                               const uint64_t n_pairs, uint4 *__restrict__ partials, uint32_t (*sm)[8], const uint32_t vbx, const uint32_t vgx) {
    const uint32_t t = c.t;
    const int32_t nv = node_value((int)t);
    const Fr tf = node_constant(nv);
    Fr acc = fr_zero();
    const uint64_t stride = (uint64_t)vgx * kBlock;
    if (c.M <= (uint32_t)kMaxFusedM) { // carry-free arithmetic (the sums come out scaled by 2^(-5(M-1)): finalize is told so)
        Fe a = fe_zero();
        uint32_t iter = 0;
        for (uint64_t b = (uint64_t)vbx * kBlock + threadIdx.x; b < n_pairs; b += stride, ++iter) {
            a = fe_carry_pass(fe_add(a, combo_product_fe(tab, c, slot_table, slot_exp, b, nv)));
            if ((iter & 31u) == 31u) a = fe_from_fr(fe_to_fr(a)); // keep the top limb far from 2^31 on long grid-stride loops
        }
        acc = fe_to_fr(a);
    } else {
        for (uint64_t b = (uint64_t)vbx * kBlock + threadIdx.x; b < n_pairs; b += stride) acc = fr_add(acc, combo_product(tab, c, slot_table, slot_exp, b, nv, tf));
    }
    const Fr s = block_sum(acc, sm);
    if (threadIdx.x == 0) fr_store(partials + 2 * (c.partial_off + (uint64_t)t * vgx + vbx), s);
}

__global__ __launch_bounds__(kBlock) void k_sum_combos(const TablePtrs tp, const Combo *__restrict__ combos,
                                                       const uint32_t *__restrict__ slot_table, const uint32_t *__restrict__ slot_exp,
                                                       const uint64_t n_pairs, uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    sum_combo_body([&](uint32_t u) { return tp.src[u]; }, combos[blockIdx.y], slot_table, slot_exp, n_pairs, partials, sm, blockIdx.x, gridDim.x);
}
// more tables than TablePtrs holds (kMaxSmallTables): the table pointers come from device memory
__global__ __launch_bounds__(kBlock) void k_sum_combos_ptrs(const uint4 *const *__restrict__ cur_tables, const Combo *__restrict__ combos,
                                                            const uint32_t *__restrict__ slot_table, const uint32_t *__restrict__ slot_exp,
                                                            const uint64_t n_pairs, uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    sum_combo_body([&](uint32_t u) { return cur_tables[u]; }, combos[blockIdx.y], slot_table, slot_exp, n_pairs, partials, sm, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(kBlock) void k_sum_combos_meta(const TablePtrs tp, const ComboMeta meta, const uint64_t n_pairs,
                                                            uint4 *__restrict__ partials) {
    __shared__ uint32_t sm[kBlock / 64][8];
    sum_combo_body([&](uint32_t u) { return tp.src[u]; }, meta.combo[blockIdx.y], meta.slot_table, meta.slot_exp, n_pairs, partials, sm, blockIdx.x, gridDim.x);
}

template <bool kLds, bool kMeta>
__global__ __launch_bounds__(kFinBlock) void k_finalize(const FinProd *__restrict__ prods, const FinMeta meta, const uint4 *__restrict__ Wm, const int K, const int D,
                                                        const int nblocks, const uint4 *__restrict__ partials, uint4 *__restrict__ scratch,
                                                        uint4 *__restrict__ out, uint64_t *__restrict__ out_wide,
                                                        uint4 *__restrict__ h_out, uint32_t *__restrict__ h_flag, const uint32_t seq,
                                                        const int scaled) {
    extern __shared__ uint4 fin_lds[];
    if constexpr (kLds) scratch = fin_lds;
    // the per-product records come from the kernel argument when they fit (kMeta): no global load in front of the partial loads
    auto prod_of = [&](int k) -> FinProd {
        if constexpr (kMeta) return meta.prod[k];
        else return prods[k];
    };
    finalize_body<kFinBlock>(prod_of, Wm, K, D, nblocks, partials, scratch, out, out_wide, h_out, h_flag, seq, scaled);
}

// The same step spread over blocks.  One block adding up 768 x 14 partials is bound by what a single CU can pull from memory
// (344 KB: ~7 us) and by the start-up of a 1024-thread block (profiles/r2d_finalize_phases.txt), so: one 256-thread block per VALID
// (product, node) pair adds up that pair's partials and leaves the sum in `sums`; the block that arrives last at the counter turns
// the sums into the message.  The hand-over is a handful of 32-byte elements, so its device-scope release/acquire is cheap (unlike
// the same protocol inside the round kernel, whose L2 is full of freshly bound table lines).  With at most eight partials per
// pair the launch is a single block.  The counter is left at zero.
constexpr int kFinMbBlock = 256;
__global__ __launch_bounds__(kFinMbBlock) void k_finalize_mb(const FinMeta meta, const uint4 *__restrict__ Wm, const int K, const int D, const int nblocks,
                                                             const uint4 *__restrict__ partials, uint4 *__restrict__ sums, uint32_t *__restrict__ counter,
                                                             uint4 *__restrict__ out, uint64_t *__restrict__ out_wide, uint4 *__restrict__ h_out,
                                                             uint32_t *__restrict__ h_flag, const uint32_t seq, const int scaled, const ClaimArgs C) {
    extern __shared__ uint4 fin_lds[];
    auto prod_of = [&](int k) -> FinProd { return meta.prod[k]; };
    if (gridDim.x == 1) {
        finalize_body<kFinMbBlock>(prod_of, Wm, K, D, nblocks, partials, fin_lds, out, out_wide, h_out, h_flag, seq, scaled);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint4 xwave[(kFinMbBlock / 64) * 2];
    __shared__ uint32_t arrived;
    const bool compact = fin_compact<kFinMbBlock>(K, D);
    Fr w_pre = fr_zero();
    if (compact) w_pre = fin_prefetch_weight(prod_of, Wm, K, D, scaled); // (every block: which one finishes the message is not known yet)
    int k = 0, t = (int)blockIdx.x; // block v -> the v-th valid (k, t)
    for (; k < K; ++k) {
        const int cnt = min((int)meta.prod[k].M, D - 1) + 1;
        if (t < cnt) break;
        t -= cnt;
    }
    const uint4 *base = partials + 2 * (meta.prod[k].partial_off + (uint64_t)t * nblocks);
    Fr acc = fr_zero();
    constexpr int kLoads = 3; // 768 partials = one batch
    // C.skip1: the round kernel left node 1 out.  This block's "sum" is then the product's CLAIM, the previous round's polynomial at
    // the challenge = sum_s lam_s(r) S_prev[s] (host-computed weights, one product per lane; the shuffles below add them up), and the
    // block that finishes the message turns it into S[1] = claim - S[0].
    const bool claim_block = C.skip1 && t == 1;
    if (claim_block) {
        const int M = (int)meta.prod[k].M;
        if ((int)threadIdx.x <= M) acc = fr_mul(fr_from_host(C.lam[claim_off(M) + (int)threadIdx.x]), fr_load(C.prev + 2 * (k * D + (int)threadIdx.x)));
    }
    for (int b0 = threadIdx.x; b0 < (claim_block ? 0 : nblocks); b0 += kFinMbBlock * kLoads) {
        Fr x[kLoads];
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int blk = b0 + kFinMbBlock * j;
            const Fr ld = fr_load(base + 2 * min(blk, nblocks - 1));
#pragma unroll
            for (int i = 0; i < 8; ++i) x[j].v[i] = blk < nblocks ? ld.v[i] : 0u;
        }
        acc = fr_add(acc, fr_add(fr_add(x[0], x[1]), x[2]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = fr_add(acc, fr_shfl_down(acc, off));
    if (lane == 0) fr_store(xwave + 2 * wave, acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kFinMbBlock / 64; ++w) acc = fr_add(acc, fr_load(xwave + 2 * w));
        fr_store(sums + 2 * (k * D + t), acc);
        arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); // releases the sum, acquires the others'
    }
    __syncthreads();
    if (arrived + 1 != gridDim.x) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // every wavefront of the last block reads other blocks' sums
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int c = threadIdx.x; c < K * D; c += kFinMbBlock) {
        if (c % D <= (int)meta.prod[c / D].M) {
            Fr v = fr_load(sums + 2 * c);
            if (C.skip1 && c % D == 1) { // claim -> node 1's sum; `sums` keeps the complete set for the next round's claims
                v = fr_sub(v, fr_load(sums + 2 * (c - 1)));
                fr_store(sums + 2 * c, v);
            }
            fr_store(fin_lds + 2 * c, v);
        }
    }
    __syncthreads();
    finalize_message<kFinMbBlock>(prod_of, Wm, K, D, fin_lds, out, out_wide, h_out, h_flag, seq, scaled, compact ? &w_pre : nullptr);
}

// ------------------------------------------------------------------------------------------------
// The persistent tail kernel: ALL latency-bound rounds of a proof (<= kSmallRoundPairs pairs) in ONE launch.
// Between dependent kernel launches the command processor needs ~5 us; a late round is three or four of them for a few
// microseconds of arithmetic.  Here a resident grid walks the rounds itself:
//   bind every table with the round's challenge        (all blocks; the lane-per-entry body of k_fix_multi)
//   -- grid barrier --
//   every (product, node) combination over the pairs   (all blocks, as virtual blocks of k_sum_combos with the same partial layout)
//   -- grid barrier --
//   block 0: finalize_body -> the message, straight into host-mapped memory, sequence flag; then it polls the host-mapped
//   word for the next challenge (bounded, as k_wait_challenge), stores it in device memory and releases the other blocks.
// The host only hashes and answers.  Grid barrier: a monotone arrival counter and a generation word, agent-scope
// release / acquire (the L2s of the eight XCDs are not coherent with each other for plain accesses; the fences write back and
// invalidate).  All blocks are co-resident by construction (at most one small block per CU).
// ------------------------------------------------------------------------------------------------
// Grid barrier over the first n_blocks blocks.  A counter that every block increments serialises at the memory side (device-scope
// atomics on one address: ~0.1 us each, 25 us for 256 blocks), so every block raises its OWN flag word instead (plain stores to
// distinct addresses), block 0's lanes watch one flag each, and the others watch the generation word block 0 then publishes.
// sync layout (uint32): [0] generation [1] -- [2] challenges released [3] stop [16 ...] one flag per block
// Every wait is BOUNDED (max_spins polls, the bound of the mailbox poll): the barrier needs all of the grid's blocks resident at once,
// which the host guarantees for the library's own launches (one tail kernel per device at a time, grid <= what the runtime says is
// co-resident) but cannot guarantee against other code of the process that occupies the GPU.  A block whose wait expires marks itself
// dead (flag 0xffffffff), tells the host (give-up marker: the host voids the proof) and leaves; block 0 turns a dead or missing block
// into the stop bit of the generation word, which releases -- and dismisses -- everybody else.  Returns false when the grid is stopping.
constexpr uint32_t kGenStop = 0x80000000u, kFlagDead = 0xffffffffu;
__device__ __forceinline__ bool grid_barrier(uint32_t *sync, uint32_t &gen, const uint32_t n_blocks, const uint32_t max_spins, uint32_t *giveup_host,
                                             const uint32_t giveup_val, uint32_t &stop_sh) {
    __syncthreads();
    gen += 1;
    if (blockIdx.x == 0) {
        bool dead = false;
        for (uint32_t f = 1 + threadIdx.x; f < n_blocks; f += kBlock) {
            uint32_t v, spins = 0;
            while ((v = __hip_atomic_load(sync + 16 + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < gen) {
                if (++spins > max_spins) { dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            dead |= v == kFlagDead;
        }
        if (dead) stop_sh = 1;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            if (stop_sh) {
                __hip_atomic_store(sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(giveup_host, giveup_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(sync, stop_sh ? (gen | kGenStop) : gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // this block's writes (every thread's: ordered by the barrier above) reach the device
        __hip_atomic_store(sync + 16 + blockIdx.x, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // relaxed polls (an acquire per poll would invalidate caches every time), one acquire fence at the end
        uint32_t v, spins = 0;
        while (((v = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~kGenStop) < gen) {
            if (++spins > max_spins) { // block 0 never came (not resident?): leave, and let it know should it ever arrive
                __hip_atomic_store(sync + 16 + blockIdx.x, kFlagDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(giveup_host, giveup_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                v = kGenStop;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (v & kGenStop) stop_sh = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return stop_sh == 0;
}

// blocks a round of `n_pairs` pairs can use: one per 256 bind outputs, and at least one per (combination, 256 pairs) tile
__device__ __forceinline__ uint32_t tail_active_blocks(const uint64_t n_pairs, const int n_tables, const int n_combos, const uint32_t G) {
    if (n_pairs <= tail_flat_pairs(n_combos)) return 1; // one block does the whole round (flat mode below)
    const uint64_t bind_blocks = (2 * n_pairs * (uint64_t)n_tables + kBlock - 1) / kBlock;
    const uint64_t sum_blocks = ((n_pairs + kBlock - 1) / kBlock) * (uint64_t)n_combos;
    return (uint32_t)min((uint64_t)G, max(bind_blocks, sum_blocks));
}

#ifdef SC_TAIL_CLOCKS // tools/build_variant.sh tail_clocks -DSC_TAIL_CLOCKS: where a tail round's time goes (100 MHz wall clock, block 0)
__device__ uint64_t g_tail_clk[64 * 8];
#define TAIL_STAMP(j, i)                                                                                                                  \
    do {                                                                                                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (j) < 64) g_tail_clk[8 * (j) + (i)] = wall_clock64();                                  \
    } while (0)
#else
#define TAIL_STAMP(j, i)
#endif
__global__ __launch_bounds__(kBlock) void k_tail_rounds(const TailArgs A, const ComboMeta meta, const FinMeta fin) {
    __shared__ uint32_t sm[kBlock / 64][8];
    __shared__ uint64_t r_sh[4];
    __shared__ uint32_t stop_sh;
    extern __shared__ uint4 fin_lds[];
    const uint32_t G = gridDim.x;
    // (the barriers' bound guards against blocks that are never scheduled, not against a slow host: never shorter than ~a second of polls,
    // however short the patience for the next challenge is)
    const uint32_t bar_spins = A.max_spins > (1u << 20) ? A.max_spins : (1u << 20);
    uint32_t gen = 0;
    uint64_t n_pairs = A.first_pairs;
    int binds = 0; // binds done so far: table u's current evaluations are cur0 (0), b0 (odd), b1 (even > 0)
    auto tab = [&](uint32_t u) -> const uint4 * { return binds == 0 ? A.t.cur0[u] : (binds & 1) ? A.t.b0[u] : A.t.b1[u]; };
    auto prod_of = [&](int k) -> FinProd { return fin.prod[k]; };
    // The round metadata and the current table pointers live in LDS: as kernel arguments, a per-lane index into them (flat mode: every
    // lane has its own combination) is a load from memory -- two of them in a row before a slot's entries can even be requested.
    __shared__ Combo combo_sh[kMetaCombos];
    __shared__ uint32_t slot_table_sh[kMetaSlots], slot_exp_sh[kMetaSlots];
    __shared__ const uint4 *tab_sh[kMaxSmallTables];
    __shared__ int prod_index_sh[kMetaCombos]; // the position (in fin.prod) of each combination's product
    for (int i = threadIdx.x; i < kMetaCombos; i += kBlock) {
        combo_sh[i] = meta.combo[i];
        int k = 0;
        if (i < A.n_combos)
            while (k < A.K - 1 && fin.prod[k].partial_off != meta.combo[i].partial_off) ++k;
        prod_index_sh[i] = k;
    }
    for (int i = threadIdx.x; i < kMetaSlots; i += kBlock) {
        slot_table_sh[i] = meta.slot_table[i];
        slot_exp_sh[i] = meta.slot_exp[i];
    }
    auto tab_lds = [&](uint32_t u) -> const uint4 * { return tab_sh[u]; };
    if (threadIdx.x == 0) stop_sh = 0;
    // (keeping block 0's Lagrange weights in registers for the whole launch was measured: eight more live registers, the first spills
    // of this kernel, no change in the round time -- the weights are L2 hits after the first round)
    const Fr *w_pre = nullptr;
    for (int j = 0; j < A.n_rounds; ++j, n_pairs >>= 1) {
        // The rounds shrink: blocks beyond what this round can use retire for good (the count never grows again), so the barriers of
        // the later rounds synchronise a handful of blocks instead of one per CU, and the last rounds run in block 0 alone.
        const uint32_t Gj = tail_active_blocks(n_pairs, A.n_tables, A.n_combos, G);
        if (blockIdx.x >= Gj) return;
        TAIL_STAMP(j, 0); // round start (block 0 has this round's challenge)
        const bool solo = Gj == 1; // no other block left: block barriers are enough
        if (j > 0 || A.first_has_bind) {
            // ---- this round's challenge: round 0's came with the launch; block 0 fetched the later ones from the host itself (below)
            // and the other blocks read them from device memory once block 0 has released them
            if (j == 0) {
                if (threadIdx.x < 4) r_sh[threadIdx.x] = A.r0.l[threadIdx.x];
            } else if (blockIdx.x != 0) {
                if (threadIdx.x == 0) {
                    // (block 0's own poll of the host is bounded by max_spins polls over PCIe, slower ones than these; four times that and a
                    // floor cover it, and a block 0 that is gone for good)
                    uint32_t spins = 0;
                    bool expired = false;
                    while (__hip_atomic_load(A.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)j) {
                        if (++spins > 16u * (A.max_spins / 4 + 1) + 4096u) { expired = true; break; } // (far beyond block 0's own patience)
                        __builtin_amdgcn_s_sleep(2);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    stop_sh = __hip_atomic_load(A.sync + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (expired) {
                        __hip_atomic_store(A.sync + 16 + blockIdx.x, kFlagDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(A.sig + 1, A.sig0 + (uint32_t)j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        stop_sh = 1;
                    }
                }
                __syncthreads();
                if (threadIdx.x < 4) r_sh[threadIdx.x] = __hip_atomic_load(A.chal + 4 * (j & 1) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (stop_sh) return; // the host never answered: block 0 has told it (give-up marker); nothing more is published
            FrHost rh;
#pragma unroll
            for (int i = 0; i < 4; ++i) rh.l[i] = r_sh[i];
            const FeU r32 = feu_shl5(fru_from_host(rh).v); // the carry-free bind's multiplier: r * 2^5 as 29-bit limbs
            // ---- bind: out[b] = in[2b] + r (in[2b+1] - in[2b]), 2 n_pairs outputs per table ------------------------------------
            const uint64_t n_out = 2 * n_pairs; // a power of two
            const int sh = 63 - __builtin_clzll(n_out);
            const uint64_t total = n_out * (uint64_t)A.n_tables;
            for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)Gj * kBlock) {
                const uint32_t u = (uint32_t)(i >> sh);
                const uint64_t b = i & (n_out - 1);
                const uint4 *src = tab(u);
                uint4 *dst = (binds & 1) ? A.t.b1[u] : A.t.b0[u];
                Fe lo, hi;
                const int32_t *stop = binds == 0 ? A.t.cur0_top[u] : nullptr;
                if (stop) { // the table arrives from the big rounds in F29, leaves canonical
                    const int2 t = *reinterpret_cast<const int2 *>(stop + 2 * b);
                    lo = fe_load_f29(src, 2 * b, t.x);
                    hi = fe_load_f29(src, 2 * b + 1, t.y);
                } else {
                    lo = fe_from_fr(fr_load(src + 4 * b));
                    hi = fe_from_fr(fr_load(src + 4 * b + 2));
                }
                fr_store(dst + 2 * b, fe_to_fr(fe_add(lo, fe_mul_u<true>(fe_sub(hi, lo), r32))));
            }
            binds += 1;
            if (solo) {
                __syncthreads();
            } else if (!grid_barrier(A.sync, gen, Gj, bar_spins, A.sig + 1, A.sig0 + (uint32_t)j + 1u, stop_sh)) {
                return;
            }
        }
        if (threadIdx.x < (uint32_t)A.n_tables) tab_sh[threadIdx.x] = tab(threadIdx.x); // where this round's tables are
        __syncthreads();
        TAIL_STAMP(j, 1); // bound (and barrier passed)
        if (solo) {
            // ---- flat mode: lane i of the block = (combination i / n_pairs, pair i % n_pairs); the pairs of a combination are
            // n_pairs adjacent lanes of one wavefront, summed by log2(n_pairs) shuffles; the sums go straight to finalize's scratch
            const int shp = 63 - __builtin_clzll(n_pairs);
            const uint32_t items = (uint32_t)A.n_combos << shp;
            for (uint32_t i0 = 0; i0 < items; i0 += kBlock) { // block-uniform trip count
                const uint32_t i = i0 + threadIdx.x;
                const bool live = i < items;
                const uint32_t ci = live ? (i >> shp) : 0;
                const Combo c = combo_sh[ci];
                const uint64_t b = i & (uint32_t)(n_pairs - 1);
                const int32_t nv = node_value((int)c.t);
                const Fr tf = node_constant(nv);
                Fr prod = fr_zero();
                if (live) {
                    if (c.M <= (uint32_t)kMaxFusedM) prod = fe_to_fr(combo_product_fe(tab_lds, c, slot_table_sh, slot_exp_sh, b, nv));
                    else prod = combo_product(tab_lds, c, slot_table_sh, slot_exp_sh, b, nv, tf);
                }
                for (uint32_t off = (uint32_t)n_pairs >> 1; off >= 1; off >>= 1) prod = fr_add(prod, fr_shfl_down(prod, (int)off));
                if (live && b == 0) fr_store(fin_lds + 2 * (prod_index_sh[ci] * A.D + (int)c.t), prod); // scratch[k * D + t]
            }
            __syncthreads();
            TAIL_STAMP(j, 2); // node sums ready
            finalize_message<kBlock>(prod_of, A.Wm, A.K, A.D, fin_lds, (uint4 *)nullptr, (uint64_t *)nullptr, A.h_out, A.h_flag, A.seq0 + (uint32_t)j, 1, w_pre);
        } else {
            // ---- sums: virtual blocks (vx, combo) of the k_sum_combos launch this round would have been -------------------------
            const uint32_t vgx = (uint32_t)((n_pairs + kBlock - 1) / kBlock);
            const uint32_t n_virtual = vgx * (uint32_t)A.n_combos;
            for (uint32_t v = blockIdx.x; v < n_virtual; v += Gj) {
                const uint32_t cy = v / vgx, vx = v % vgx;
                sum_combo_body(tab_lds, combo_sh[cy], slot_table_sh, slot_exp_sh, n_pairs, A.partials, sm, vx, vgx);
            }
            if (!grid_barrier(A.sync, gen, Gj, bar_spins, A.sig + 1, A.sig0 + (uint32_t)j + 1u, stop_sh)) return;
            TAIL_STAMP(j, 2); // partial sums ready (barrier passed)
            // (vgx <= kTailMaxPairs / kBlock = 8 partials per combination: block 0 adds them up itself)
            if (blockIdx.x == 0) {
                finalize_body<kBlock>(prod_of, A.Wm, A.K, A.D, (int)vgx, A.partials, fin_lds, (uint4 *)nullptr, (uint64_t *)nullptr, A.h_out, A.h_flag,
                                      A.seq0 + (uint32_t)j, 1, w_pre);
            }
        }
        // ---- block 0: the next challenge.  The host stores, for limb i of the challenge, the 64-bit word (limb << 32 | tag) into
        // slot word i, tag = the low 32 bits of sig0 + j + 1 xor-ed into nothing else: eight lanes poll one word each until its
        // tag matches, so the challenge arrives with the poll that sees it (no second trip over PCIe to fetch it).
        TAIL_STAMP(j, 3); // message published
        if (blockIdx.x == 0 && j + 1 < A.n_rounds) {
            const uint32_t want = A.sig0 + (uint32_t)(j + 1);
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                uint64_t w = 0;
                bool seen = false;
                for (uint32_t spin = 0; spin < A.max_spins; ++spin) {
                    if (lane < 8) w = __hip_atomic_load(A.mail_host + 8 * (want & 1u) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const bool mine = lane >= 8 || (uint32_t)w == want;
                    if (__all(mine)) { seen = true; break; }
                    // word 0 tagged with the stop bit: the host asks the kernel to leave (the interactive protocol's resident kernel is
                    // quiesced before any other entry point touches the handle) -- same clean exit as an expired wait
                    if (__any(lane == 0 && (uint32_t)w == (want ^ 0x80000000u))) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!seen && lane == 0) { // gave up: tell the host (it voids the proof) and let every block leave
                    __hip_atomic_store(A.sig + 1, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(A.sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stop_sh = 1;
                }
                // limbs 2i, 2i+1 (lanes 2i, 2i+1) -> 64-bit limb i
                const uint32_t lo32 = (uint32_t)(w >> 32);
                const uint32_t hi32 = __shfl_down(lo32, 1, 64);
                if (lane < 8 && (lane & 1) == 0) {
                    const uint64_t limb = (uint64_t)lo32 | ((uint64_t)hi32 << 32);
                    r_sh[lane >> 1] = limb;
                    __hip_atomic_store(A.chal + 4 * ((j + 1) & 1) + (lane >> 1), limb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lane == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_store(A.sync + 2, (uint32_t)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                TAIL_STAMP(j, 4); // next challenge fetched and released
            }
        }
    }
}

// how many blocks of k_tail_rounds the device holds at once (its barriers need every launched block resident): asked of the
// runtime, for the worst-case dynamic LDS, once per process and device
int tail_max_resident_blocks(int device) {
    static int cached_dev = -1, cached = 0;
    if (cached_dev == device) return cached;
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tail_rounds, kBlock, kFinLdsMax) != hipSuccess || per_cu < 1 ||
        hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount < 1) {
        (void)hipGetLastError();
        return 0;
    }
    cached = std::min(per_cu * prop.multiProcessorCount, kTailMaxGrid);
    cached_dev = device;
    return cached;
}

// (two regions in one launch: k_tail_slices' synchronisation words and its accumulator ring -- a second launch is 5-8 us on the critical path)
__global__ void k_zero_words(uint32_t *p, const uint32_t n, uint32_t *p2, const uint32_t n2) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x) p2[i] = 0u;
}
hipError_t launch_zero_words(uint32_t *p, uint32_t n, hipStream_t stream, uint32_t *p2, uint32_t n2) {
    hipLaunchKernelGGL(k_zero_words, dim3(4), dim3(256), 0, stream, p, n, p2, p2 ? n2 : 0u);
    return hipGetLastError();
}

// W[i] = W0[i] * (lo + r (hi - lo)) for a bound two-entry table {lo, hi}: a handle whose polynomial is multiplied by a table's value at the
// point the rounds fixed (GKR phase two: f2(u) as the product's coefficient, gkr_round_sumcheck/mod.rs:71-75,122) without visiting the host
__global__ void k_scale_w_by_table_eval(uint4 *__restrict__ W, const uint4 *__restrict__ W0, const uint32_t n, const uint4 *__restrict__ table, const FrU r) {
    const Fr lo = fr_load(table), hi = fr_load(table + 2);
    const Fr s = fr_add(lo, fr_mul_u(fr_sub(hi, lo), r));
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) fr_store(W + 2 * (size_t)i, fr_mul(fr_load(W0 + 2 * (size_t)i), s));
}
hipError_t launch_scale_w_by_table_eval(FrHost *W, const FrHost *W0, uint32_t n, const void *table, const FrHost &r, hipStream_t stream) {
    FrU ru;
    for (int i = 0; i < 4; ++i) {
        ru.v[2 * i] = (uint32_t)r.l[i];
        ru.v[2 * i + 1] = (uint32_t)(r.l[i] >> 32);
    }
    hipLaunchKernelGGL(k_scale_w_by_table_eval, dim3(1), dim3(64), 0, stream, reinterpret_cast<uint4 *>(W), reinterpret_cast<const uint4 *>(W0), n,
                       static_cast<const uint4 *>(table), ru);
    return hipGetLastError();
}

hipError_t launch_tail_rounds(const TailArgs &args, const ComboMeta &meta, const FinMeta &fin, int grid, hipStream_t stream) {
    const size_t lds = (size_t)args.K * args.D * (args.D + 2) * 32;
    if (lds > kFinLdsMax) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tail_rounds, dim3(grid), dim3(kBlock), lds, stream, args, meta, fin);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Synthetic uniform field elements: SplitMix64 keyed by (seed, stream, index); the same
// specification as oracle/oracle.c:orc_synth_table and oracle/pyoracle.py:synth_mont_limbs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void k_synth(const uint64_t key, const uint64_t first, const uint64_t n, uint4 *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t index = first + i;
        for (uint64_t attempt = 0; attempt < 64; ++attempt) { // P(reject) = 0.094 per attempt
            uint64_t l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) l[k] = splitmix64(key ^ (splitmix64(index * 4 + k) + attempt * 0x9E3779B97F4A7C15ULL));
            l[3] &= 0xffffffffffffffffULL >> 1;
            // accept iff < p
            const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
            bool lt = false, decided = false;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                if (!decided && l[k] != P[k]) { lt = l[k] < P[k]; decided = true; }
            }
            if (lt) {
                out[2 * i] = make_uint4((uint32_t)l[0], (uint32_t)(l[0] >> 32), (uint32_t)l[1], (uint32_t)(l[1] >> 32));
                out[2 * i + 1] = make_uint4((uint32_t)l[2], (uint32_t)(l[2] >> 32), (uint32_t)l[3], (uint32_t)(l[3] >> 32));
                break;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// modmul ceiling micro-kernel: a dependent chain of `reps` Montgomery products per lane, 4 independent
// chains per lane for ILP.  variant 0: CIOS product; 1: fr_add chain (VALU add/sub ceiling); 2: Comba product; 3: Comba by a
// wave-uniform operand; 4: two Comba products interleaved per asm statement.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_bench_modmul(const uint32_t reps, const uint32_t variant, uint64_t *__restrict__ sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    Fr x[4], y;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[c].v[i] = (uint32_t)splitmix64(gid * 64 + c * 8 + i) & (i == 7 ? 0x3fffffffu : 0xffffffffu);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) y.v[i] = (uint32_t)splitmix64(~gid * 8 + i) & (i == 7 ? 0x3fffffffu : 0xffffffffu);
    if (variant == 0) {
        for (uint32_t k = 0; k < reps; ++k) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = fr_mul_cios(x[c], y);
        }
    } else if (variant == 2) {
        for (uint32_t k = 0; k < reps; ++k) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = fr_mul_comba(x[c], y);
        }
    } else if (variant == 4) {
        for (uint32_t k = 0; k < reps; ++k) {
            fr_mul2_comba(x[0], y, x[1], y, x[0], x[1]);
            fr_mul2_comba(x[2], y, x[3], y, x[2], x[3]);
        }
    } else if (variant == 3) {
        FrU u;
#pragma unroll
        for (int i = 0; i < 8; ++i) u.v[i] = __builtin_amdgcn_readfirstlane(y.v[i]);
        for (uint32_t k = 0; k < reps; ++k) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = fr_mul_comba_u(x[c], u);
        }
    } else {
        for (uint32_t k = 0; k < reps; ++k) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = fr_add(x[c], y);
        }
    }
    uint64_t h = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) h = h * 0x100000001B3ULL + x[c].v[i];
    }
    if (h == 0x1234567ULL || gid == 0) sink[0] = h; // keep the chain alive
}

// ------------------------------------------------------------------------------------------------
// elementwise field ops on arrays (parity tests of the arithmetic itself, every implementation)
//   op 0 mul (production) | 1 add | 2 sub | 3 mul CIOS (plain C++) | 4 mul Comba (asm) | 5 mul by uniform u
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_fr_elementwise(const int op, const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                                           const FrHost u_h, uint4 *__restrict__ out, const uint64_t n) {
    const FrU u = fru_from_host(u_h);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const Fr x = fr_load(a + 2 * i), y = fr_load(b + 2 * i);
        Fr z;
        switch (op) {
        case 0: z = fr_mul(x, y); break;
        case 1: z = fr_add(x, y); break;
        case 2: z = fr_sub(x, y); break;
        case 3: z = fr_mul_cios(x, y); break;
        case 4: z = fr_mul_comba(x, y); break;
        default: z = fr_mul_comba_u(x, u); break;
        }
        fr_store(out + 2 * i, z);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int grid_cap() {
#ifdef SC_EXPERIMENTS // SC_GRID: at most kMaxGrid (the partial buffers are sized for that)
    static const int cap = [] {
        const char *e = std::getenv("SC_GRID");
        const int v = e ? std::atoi(e) : kMaxGrid;
        return v >= 1 && v <= kMaxGrid ? v : kMaxGrid;
    }();
    return cap;
#else
    return kMaxGrid;
#endif
}
int grid_for_pairs(uint64_t n_pairs) {
    uint64_t g = (n_pairs + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > (uint64_t)grid_cap()) g = grid_cap();
    return (int)g;
}

hipError_t launch_sum_generic(const uint4 *const *d_cur_tables, const uint32_t *d_slot_table, const uint32_t *d_slot_exp,
                              int n_slots, int M, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_sum_generic, dim3(grid, M + 1), dim3(kBlock), 0, stream, d_cur_tables, d_slot_table, d_slot_exp, n_slots, M,
                       n_pairs, (uint4 *)d_partials);
    return hipGetLastError();
}

hipError_t launch_fix(const uint4 *src, uint4 *dst, const FrHost &r, uint64_t n_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_fix, dim3(grid_for_pairs(n_out)), dim3(kBlock), 0, stream, src, dst, r, n_out);
    return hipGetLastError();
}

hipError_t launch_scale(const uint4 *src, uint4 *dst, const FrHost &s, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_scale, dim3(grid_for_pairs(n)), dim3(kBlock), 0, stream, src, dst, s, n);
    return hipGetLastError();
}

// the multi-block form with more than one block: every (product, node) sum of the round is left in d_scratch
bool finalize_keeps_sums(int K, int D, int nblocks, bool have_host_prods, bool have_counter) {
    return (size_t)K * D * (D + 2) * 32 <= kFinLdsMax && have_host_prods && K <= kMetaProds && have_counter && nblocks > 8;
}

hipError_t launch_finalize(const FinProd *d_prods, const FinProd *h_prods_or_null, const FrHost *d_W, int K, int D, int nblocks, const FrHost *d_partials, FrHost *d_scratch,
                           FrHost *d_out, uint64_t *d_out_wide, FrHost *h_out_mapped, uint32_t *h_flag_mapped, uint32_t seq,
                           int scaled, uint32_t *d_counter_or_null, hipStream_t stream, const ClaimArgs *claim_or_null) {
    const size_t lds = (size_t)K * D * (D + 2) * 32;
    ClaimArgs claim;
    std::memset(&claim, 0, sizeof(claim));
    if (claim_or_null) claim = *claim_or_null;
    if (claim.skip1 && !finalize_keeps_sums(K, D, nblocks, h_prods_or_null != nullptr, d_counter_or_null != nullptr)) return hipErrorInvalidValue;
    FinMeta meta;
    std::memset(&meta, 0, sizeof(meta));
    const bool use_meta = h_prods_or_null && K <= kMetaProds;
    if (use_meta) std::memcpy(meta.prod, h_prods_or_null, (size_t)K * sizeof(FinProd));
    plan_hit(lds <= kFinLdsMax && use_meta && d_counter_or_null ? kPlanFinalizeMultiBlock : lds <= kFinLdsMax ? kPlanFinalizeOneBlock : kPlanFinalizeNoLds);
    if (lds <= kFinLdsMax && use_meta && d_counter_or_null) {
        int n_valid = 0;
        for (int k = 0; k < K; ++k) n_valid += std::min<int>((int)meta.prod[k].M, D - 1) + 1;
        hipLaunchKernelGGL(k_finalize_mb, dim3(nblocks <= 8 ? 1 : n_valid), dim3(kFinMbBlock), lds, stream, meta, (const uint4 *)d_W, K, D, nblocks,
                           (const uint4 *)d_partials, (uint4 *)d_scratch, d_counter_or_null, (uint4 *)d_out, d_out_wide, (uint4 *)h_out_mapped,
                           h_flag_mapped, seq, scaled, claim);
    } else if (lds <= kFinLdsMax && use_meta)
        hipLaunchKernelGGL((k_finalize<true, true>), dim3(1), dim3(kFinBlock), lds, stream, d_prods, meta, (const uint4 *)d_W, K, D, nblocks,
                           (const uint4 *)d_partials, (uint4 *)d_scratch, (uint4 *)d_out, d_out_wide, (uint4 *)h_out_mapped, h_flag_mapped, seq,
                           scaled);
    else if (lds <= kFinLdsMax)
        hipLaunchKernelGGL((k_finalize<true, false>), dim3(1), dim3(kFinBlock), lds, stream, d_prods, meta, (const uint4 *)d_W, K, D, nblocks,
                           (const uint4 *)d_partials, (uint4 *)d_scratch, (uint4 *)d_out, d_out_wide, (uint4 *)h_out_mapped, h_flag_mapped, seq,
                           scaled);
    else
        hipLaunchKernelGGL((k_finalize<false, false>), dim3(1), dim3(kFinBlock), 0, stream, d_prods, meta, (const uint4 *)d_W, K, D, nblocks,
                           (const uint4 *)d_partials, (uint4 *)d_scratch, (uint4 *)d_out, d_out_wide, (uint4 *)h_out_mapped, h_flag_mapped, seq,
                           scaled);
    return hipGetLastError();
}

hipError_t launch_fix_multi(const TablePtrs &tp, int n_tables, const FrHost &r, const FrHost *r_mail, uint64_t n_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_fix_multi, dim3(grid_for_pairs(n_out), n_tables), dim3(kBlock), 0, stream, tp, r, r_mail, n_out);
    return hipGetLastError();
}

hipError_t launch_sum_combos(const TablePtrs &tp, const Combo *d_combos, int n_combos, const uint32_t *d_slot_table,
                             const uint32_t *d_slot_exp, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_sum_combos, dim3(grid, n_combos), dim3(kBlock), 0, stream, tp, d_combos, d_slot_table, d_slot_exp, n_pairs,
                       (uint4 *)d_partials);
    return hipGetLastError();
}

hipError_t launch_sum_combos_ptrs(const uint4 *const *d_cur_tables, const Combo *d_combos, int n_combos, const uint32_t *d_slot_table,
                                  const uint32_t *d_slot_exp, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_sum_combos_ptrs, dim3(grid, n_combos), dim3(kBlock), 0, stream, d_cur_tables, d_combos, d_slot_table, d_slot_exp, n_pairs,
                       (uint4 *)d_partials);
    return hipGetLastError();
}

hipError_t launch_sum_combos_meta(const TablePtrs &tp, const ComboMeta &meta, int n_combos, uint64_t n_pairs, FrHost *d_partials, int grid,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(k_sum_combos_meta, dim3(grid, n_combos), dim3(kBlock), 0, stream, tp, meta, n_pairs, (uint4 *)d_partials);
    return hipGetLastError();
}

hipError_t launch_synth(uint64_t seed, uint64_t stream_id, uint64_t first, uint64_t n, uint4 *d_out, hipStream_t stream) {
    // host-side SplitMix64 for the key (same function as on the device)
    uint64_t x = seed ^ (stream_id * 0xD1342543DE82EF95ULL);
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    const uint64_t key = z ^ (z >> 31);
    hipLaunchKernelGGL(k_synth, dim3(grid_for_pairs(n)), dim3(kBlock), 0, stream, key, first, n, d_out);
    return hipGetLastError();
}

// OR a generation's tag into a round's lanes (streamed rounds under an RCCL communicator with direct publication: their message is added up
// over the chunks by k_msg_accumulate, which knows nothing of tags)
__global__ void k_tag_words(uint64_t *__restrict__ w, const int n, const uint64_t tag) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) w[i] |= tag;
}
hipError_t launch_tag_words(uint64_t *d_words, int n, uint32_t gen, hipStream_t stream) {
    hipLaunchKernelGGL(k_tag_words, dim3(1), dim3(64), 0, stream, d_words, n, (uint64_t)wide_tag_of(gen) << kWideTagShift);
    return hipGetLastError();
}
// copy a few words to host-mapped memory and raise the sequence flag (after an all-reduce on the same stream)
__global__ void k_publish_words(const uint64_t *__restrict__ src, uint64_t *__restrict__ h_dst, const int n, uint32_t *__restrict__ h_flag,
                                const uint32_t seq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) h_dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_publish_words(const uint64_t *d_src, uint64_t *h_dst_mapped, int n, uint32_t *h_flag_mapped, uint32_t seq, hipStream_t stream) {
    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, stream, d_src, h_dst_mapped, n, h_flag_mapped, seq);
    return hipGetLastError();
}

// sc_comm_init_p2p: the round's all-reduce as ONE small kernel per rank (kernels.h: P2PArgs)
__global__ __launch_bounds__(kP2PWords) void k_p2p_allreduce(const P2PArgs A, uint64_t *__restrict__ lanes, uint64_t *__restrict__ h_dst,
                                                             uint32_t *__restrict__ h_flag, const uint32_t seq) {
    __shared__ uint32_t missing;
    const int w = threadIdx.x;
    const uint64_t tag = (uint64_t)(A.gen & 0xffffffu) << 40, mask = (1ULL << 40) - 1;
    const size_t half = (size_t)(A.gen & 1u) * kP2PMaxRanks * kP2PWords;
    if (w == 0) missing = 0;
    __syncthreads();
    uint64_t sum = 0;
    if (w < A.n_words) {
        const uint64_t mine = lanes[w];
        for (int q = 0; q < A.nranks; ++q) // push: my word w into slot `rank` of every inbox (my own included: one code path)
            __hip_atomic_store(A.inbox[q] + half + (size_t)A.rank * kP2PWords + w, tag | (mine & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t *in = A.inbox[A.rank] + half;
        for (int q = 0; q < A.nranks; ++q) { // pull from local memory: the poll that sees the generation has fetched the value
            uint64_t x = 0;
            uint32_t spins = 0;
            while (((x = __hip_atomic_load(in + (size_t)q * kP2PWords + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) & ~mask) != tag) {
                if (++spins > A.max_spins) {
                    missing = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            sum += x & mask;
        }
    }
    __syncthreads();
    if (missing) { // nothing is published but the request to run again (pushes are idempotent; what has arrived stays)
        if (w == 0) __hip_atomic_store(h_flag, seq | kP2PRetryBit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (w < A.n_words) {
        lanes[w] = sum;
        h_dst[w] = sum;
    }
    __threadfence_system();
    __syncthreads();
    if (w == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_p2p_allreduce(const P2PArgs &args, uint64_t *d_lanes, uint64_t *h_dst_mapped, uint32_t *h_flag_mapped, uint32_t seq, hipStream_t stream) {
    if (args.n_words > kP2PWords || args.nranks > kP2PMaxRanks) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(kP2PWords), 0, stream, args, d_lanes, h_dst_mapped, h_flag_mapped, seq);
    return hipGetLastError();
}

// table-sized all-reduce of a p2p group (sharded GKR initialisation): every rank adds up all ranks' lanes, read in place over xGMI
__global__ __launch_bounds__(kBlock) void k_sum_peer_lanes(const PeerLanes P, const uint64_t n_words, uint64_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * kBlock) {
        uint64_t s = 0;
        for (int q = 0; q < P.n; ++q) s += P.p[q][i];
        out[i] = s;
    }
}
hipError_t launch_sum_peer_lanes(const PeerLanes &peers, uint64_t n_words, uint64_t *out, hipStream_t stream) {
    const int grid = (int)std::min<uint64_t>((n_words + kBlock - 1) / kBlock, 2048);
    hipLaunchKernelGGL(k_sum_peer_lanes, dim3(std::max(grid, 1)), dim3(kBlock), 0, stream, peers, n_words, out);
    return hipGetLastError();
}

// tail of a sharded proof: recv[g][u][e] (rank-major, as an all-gather leaves it; e < per) -> tabs[u][g * per + e]: one table of
// G * per entries per polynomial, rank g's slice at the high index bits where its shard of the original table was
__global__ void k_gather_to_tables(const uint4 *__restrict__ recv, uint4 *__restrict__ tabs, const uint32_t G, const uint32_t U, const uint32_t per) {
    const uint64_t n = (uint64_t)G * U * per;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { // output element index
        const uint32_t e = (uint32_t)(i % per);
        const uint64_t ug = i / per;
        const uint32_t g = (uint32_t)(ug % G), u = (uint32_t)(ug / G);
        const uint64_t src = ((uint64_t)g * U + u) * per + e;
        tabs[2 * i] = recv[2 * src];
        tabs[2 * i + 1] = recv[2 * src + 1];
    }
}
hipError_t launch_gather_to_tables(const uint4 *recv, uint4 *tabs, uint32_t G, uint32_t U, uint32_t per, hipStream_t stream) {
    const uint64_t n = (uint64_t)G * U * per;
    hipLaunchKernelGGL(k_gather_to_tables, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, recv, tabs, G, U, per);
    return hipGetLastError();
}

// streamed tables: a round is computed chunk by chunk; every chunk's message (D elements) is added into `acc`, the last one publishes
__global__ void k_msg_accumulate(const uint4 *__restrict__ in, uint4 *__restrict__ acc, const int D, const int first, const int last,
                                 uint4 *__restrict__ d_out, uint64_t *__restrict__ out_wide, uint4 *__restrict__ h_out, uint32_t *__restrict__ h_flag,
                                 const uint32_t seq) {
    const int t = threadIdx.x;
    if (t < D) {
        Fr a = fr_load(in + 2 * t);
        if (!first) a = fr_add(a, fr_load(acc + 2 * t));
        fr_store(acc + 2 * t, a);
        if (last) {
            if (d_out) fr_store(d_out + 2 * t, a);
            if (h_out) fr_store(h_out + 2 * t, a);
            if (out_wide) // a sharded round's lanes: the eight 32-bit limbs zero-extended (summable across ranks)
                for (int j = 0; j < 8; ++j) out_wide[8 * t + j] = (uint64_t)a.v[j];
        }
    }
    if (last && h_flag) {
        __threadfence_system();
        __syncthreads();
        if (t == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
hipError_t launch_msg_accumulate(const FrHost *in, FrHost *acc, int D, bool first, bool last, FrHost *d_out, uint64_t *d_out_wide, FrHost *h_out_mapped,
                                 uint32_t *h_flag_mapped, uint32_t seq, hipStream_t stream) {
    hipLaunchKernelGGL(k_msg_accumulate, dim3(1), dim3(64), 0, stream, (const uint4 *)in, (uint4 *)acc, D, first ? 1 : 0, last ? 1 : 0, (uint4 *)d_out,
                       d_out_wide, (uint4 *)h_out_mapped, h_flag_mapped, seq);
    return hipGetLastError();
}

hipError_t launch_f29_to_sat(const uint4 *src, const int32_t *src_top, uint4 *dst, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_f29_to_sat, dim3(grid_for_pairs(n)), dim3(kBlock), 0, stream, src, src_top, dst, n);
    return hipGetLastError();
}

hipError_t launch_fr_elementwise(int op, const uint4 *a, const uint4 *b, const FrHost &u, uint4 *out, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_fr_elementwise, dim3(grid_for_pairs(n)), dim3(kBlock), 0, stream, op, a, b, u, out, n);
    return hipGetLastError();
}

hipError_t launch_bench_modmul(uint64_t n_threads, uint32_t reps, uint32_t variant, uint64_t *d_sink, hipStream_t stream) {
    hipLaunchKernelGGL(k_bench_modmul, dim3((unsigned)((n_threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, reps, variant, d_sink);
    return hipGetLastError();
}

} // namespace scd

#ifdef SC_TAIL_CLOCKS
extern "C" __attribute__((visibility("default"))) int sc_debug_tail_clocks(uint64_t *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scd::g_tail_clk), sizeof(uint64_t) * 64 * 8);
}
#endif
#ifdef SC_FIN_CLOCKS
extern "C" __attribute__((visibility("default"))) int sc_debug_fin_clocks(uint64_t *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scd::g_fin_clk), sizeof(uint64_t) * 12);
}
#endif
