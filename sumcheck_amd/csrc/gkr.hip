// gkr.hip -- GKR round sumcheck (Libra section 3.3) initialisation on the GPU and the two-phase driver.
//
// Replaces, on the device:
//   SparseMultilinearExtension::fix_variables (ark-poly; called at reference src/gkr_round_sumcheck/mod.rs:31,62)
//   initialize_phase_one  (mod.rs:22-42)   h_g[x] = sum_y f1(g,x,y) * f3[y]
//   initialize_phase_two  (mod.rs:57-63)   f1(g,u,.) as a dense table
//   start_phase2_sumcheck's f3 * f2(u)      (mod.rs:71-75) and f2.evaluate(u) (mod.rs:122)
//   GKRRoundSumcheck::prove (mod.rs:93-139) driver: both sumcheck phases over tables that never leave HBM.
//
// The reference folds sparse entries through hash maps; there is no atomic field addition on a GPU, so the
// fold is restated as  sort by index (once) -> multiply by the eq table -> segmented field sum of equal keys
// (rocPRIM reduce_by_key with the modular-add functor).  Binding the LOW k variables maps index i to key i >> k,
// which keeps a sorted list sorted, so only the scatter by x in phase one needs a second sort.  Field
// arithmetic is exact, hence the result is the same canonical table whatever the summation order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>

#include "../../include/sumcheck_hip.h"
#include "fr_device.hpp"
#include "host_fr.hpp"
#include "kernels.h"
#include "transcript.hpp"

using scd::Fr;
using scd::FrHost;
using scd::FrU;
using scd::kBlock;

int sc_internal_fail(int code, const char *fmt, ...); // abi.hip
void sc_internal_gate_lock(int device);                // abi.hip: the device gate (serialises the library's HIP calls per device)
void sc_internal_gate_unlock(int device);
int sc_internal_device();                             // abi.hip: the calling thread's device (sc_set_device)
uint64_t sc_internal_cache_limit();                   // abi.hip: sc_set_cache_limit
void sc_internal_release_eval_cache();                // abi.hip: sc_poly_evaluate's cached work areas
void sc_internal_release_handle_pool();               // abi.hip: the prover sc_ml_prove keeps between one-shot proofs
int sc_internal_run_rounds(sc_prover *p, sch::Blake2b512Rng &rng, uint32_t n_rounds, uint64_t *out_msgs, sch::Fr *out_challenges); // api.hip
hipStream_t sc_internal_prover_stream(sc_prover *p);                                                  // abi.hip (GKR phase two: see there)
const void *sc_internal_bound_table(sc_prover *p, uint32_t u);
int sc_internal_scale_by_bound_table(sc_prover *p, const void *table, const sch::Fr &r_last);
struct sc_rng {
    sch::Blake2b512Rng rng;
};

namespace {

struct GkrGate { // RAII: the calling thread's device
    int dev;
    GkrGate() : dev(sc_internal_device()) { sc_internal_gate_lock(dev); }
    ~GkrGate() { sc_internal_gate_unlock(dev); }
};

struct FrAdd {
    __device__ Fr operator()(const Fr &a, const Fr &b) const { return scd::fr_add(a, b); }
};

__device__ __forceinline__ FrU fru(const FrHost &h) {
    FrU r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.v[2 * i] = (uint32_t)h.l[i];
        r.v[2 * i + 1] = (uint32_t)(h.l[i] >> 32);
    }
    return r;
}
__device__ __forceinline__ Fr fr_ld(const Fr *p) { return scd::fr_load(reinterpret_cast<const uint4 *>(p)); }
__device__ __forceinline__ void fr_st(Fr *p, const Fr &a) { scd::fr_store(reinterpret_cast<uint4 *>(p), a); }

// precompute_eq (ark-poly): eq[b] = prod_i (b_i ? g_i : 1 - g_i), built there by doubling: dp[b + 2^i] = dp[b] * g_i ; dp[b] -= dp[b + 2^i].
// Exact field arithmetic, so any evaluation order gives the same canonical table.  Here: the low kl <= 11 variables and the high
// kh <= 11 variables are two small tables (EqSplit; the full table, where one is wanted, is their outer product: k_eq_outer).  A half is
// built in LOGARITHMIC depth -- one dependent Montgomery product is ~1 us for a lone wavefront, and the doubling recurrence is a chain
// of kl of them (16 us at kl = 10, measured): start from the kl one-variable tables {1 - g_i, g_i}, merge neighbours pairwise by outer
// product (low variables vary fastest) until two tables are left, and write their outer product -- the half -- straight to memory,
// a slice of it per block: 10 variables -> 5 -> 3 -> 2 tables -> out: four products deep.
constexpr int kEqMaxVars = 22, kEqHalfMax = 11, kEqBlock = 512, kEqSlices = 4, kEqLdsEntries = 320; // (the two last tables: 2^8 + 2^3 entries at most, a level before: less)
struct EqPoint {
    FrHost g[kEqMaxVars];
};
__device__ void eq_half_block(Fr *__restrict__ out, const EqPoint &P, const int first, const int n, const int slice, const int n_slices) { // n >= 1 variables
    __shared__ uint4 tab[2][2 * kEqLdsEntries];
    __shared__ int off[2][kEqHalfMax + 1], lg[2][kEqHalfMax + 1];
    Fr *cur = reinterpret_cast<Fr *>(tab[0]), *nxt = reinterpret_cast<Fr *>(tab[1]);
    int nt = n, w = 0;
    if ((int)threadIdx.x < n) { // level 0: {1 - g, g} per variable
        Fr g;
        const FrU gu = fru(P.g[first + threadIdx.x]);
#pragma unroll
        for (int i = 0; i < 8; ++i) g.v[i] = gu.v[i];
        fr_st(cur + 2 * threadIdx.x, scd::fr_sub(scd::fr_one(), g));
        fr_st(cur + 2 * threadIdx.x + 1, g);
        off[0][threadIdx.x] = 2 * threadIdx.x;
        lg[0][threadIdx.x] = 1;
    }
    __syncthreads();
    while (nt > 2) { // merge tables 2t and 2t + 1 (an odd one out is copied)
        const int nn = (nt + 1) / 2;
        if (threadIdx.x == 0) {
            int o = 0;
            for (int t = 0; t < nn; ++t) {
                const int l = lg[w][2 * t] + (2 * t + 1 < nt ? lg[w][2 * t + 1] : 0);
                off[w ^ 1][t] = o;
                lg[w ^ 1][t] = l;
                o += 1 << l;
            }
        }
        __syncthreads();
        for (int t = 0; t < nn; ++t) {
            const int la = lg[w][2 * t], oa = off[w][2 * t], oo = off[w ^ 1][t];
            if (2 * t + 1 < nt) {
                const int ob = off[w][2 * t + 1], sz = 1 << lg[w ^ 1][t];
                for (int e = threadIdx.x; e < sz; e += blockDim.x)
                    fr_st(nxt + oo + e, scd::fr_mul(fr_ld(cur + oa + (e & ((1 << la) - 1))), fr_ld(cur + ob + (e >> la))));
            } else {
                for (int e = threadIdx.x; e < (1 << la); e += blockDim.x) fr_st(nxt + oo + e, fr_ld(cur + oa + e));
            }
        }
        __syncthreads();
        Fr *tmp = cur;
        cur = nxt;
        nxt = tmp;
        w ^= 1;
        nt = nn;
    }
    const int total = 1 << n, per = (total + n_slices - 1) / n_slices, e0 = slice * per, e1 = min(total, e0 + per);
    if (nt == 1) {
        for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) fr_st(out + e, fr_ld(cur + off[w][0] + e));
    } else {
        const int la = lg[w][0], oa = off[w][0], ob = off[w][1];
        for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) fr_st(out + e, scd::fr_mul(fr_ld(cur + oa + (e & ((1 << la) - 1))), fr_ld(cur + ob + (e >> la))));
    }
}
// blockIdx.y = 0: the low kl variables -> lo (2^kl entries); blockIdx.y = 1 (if kh > 0): the next kh variables -> hi; blockIdx.x: slice of the half
__global__ __launch_bounds__(kEqBlock) void k_eq_halves(Fr *lo, Fr *hi, const EqPoint P, const int kl, const int kh) {
    if (blockIdx.y == 0) eq_half_block(lo, P, 0, kl, blockIdx.x, gridDim.x);
    else eq_half_block(hi, P, kl, kh, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(kBlock) void k_eq_outer(const Fr *__restrict__ lo, const Fr *__restrict__ hi, const int kl, const uint64_t n, Fr *__restrict__ eq) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t mask = (1ULL << kl) - 1;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n; b += stride) fr_st(eq + b, scd::fr_mul(fr_ld(hi + (b >> kl)), fr_ld(lo + (b & mask))));
}
// w[i] = eq[idx[i] & mask] * vals[perm ? perm[i] : i] ; key[i] = idx[i] >> k
__global__ __launch_bounds__(kBlock) void k_sparse_scale(const uint64_t *__restrict__ idx, const Fr *__restrict__ vals,
                                                         const uint32_t *__restrict__ perm, const Fr *__restrict__ eq, const uint32_t k,
                                                         const uint64_t n, uint64_t *__restrict__ key, Fr *__restrict__ w) {
    const uint64_t mask = (k >= 64) ? ~0ULL : ((1ULL << k) - 1);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t id = idx[i];
        const Fr v = fr_ld(vals + (perm ? perm[i] : i));
        fr_st(w + i, scd::fr_mul(fr_ld(eq + (id & mask)), v));
        key[i] = id >> k;
    }
}
// eq(point, idx & mask) evaluated directly (no 2^k table): used when 2^k entries would not be worth building
__global__ __launch_bounds__(kBlock) void k_sparse_scale_direct(const uint64_t *__restrict__ idx, const Fr *__restrict__ vals,
                                                                const uint32_t *__restrict__ perm, const Fr *__restrict__ point,
                                                                const uint32_t k, const uint64_t n, uint64_t *__restrict__ key,
                                                                Fr *__restrict__ w) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t id = idx[i];
        Fr acc = fr_ld(vals + (perm ? perm[i] : i));
        for (uint32_t j = 0; j < k; ++j) {
            const Fr gj = fr_ld(point + j);
            acc = scd::fr_mul(acc, ((id >> j) & 1) ? gj : scd::fr_sub(scd::fr_one(), gj));
        }
        fr_st(w + i, acc);
        key[i] = (k >= 64) ? 0 : (id >> k);
    }
}
__global__ __launch_bounds__(kBlock) void k_iota(uint32_t *p, const uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)i;
}
// phase one scatter terms (mod.rs:32-38): key_x = xy & mask, t = v * f3[xy >> dim]
__global__ __launch_bounds__(kBlock) void k_hg_terms(const uint64_t *__restrict__ xy, const Fr *__restrict__ v, const Fr *__restrict__ f3,
                                                     const uint32_t dim, const uint64_t n, uint64_t *__restrict__ key_x,
                                                     Fr *__restrict__ t) {
    const uint64_t mask = (1ULL << dim) - 1;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t id = xy[i];
        key_x[i] = id & mask;
        fr_st(t + i, scd::fr_mul(fr_ld(v + i), fr_ld(f3 + (id >> dim))));
    }
}
__global__ __launch_bounds__(kBlock) void k_gather(const Fr *__restrict__ src, const uint32_t *__restrict__ perm, const uint64_t n,
                                                   Fr *__restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) fr_st(dst + i, fr_ld(src + perm[i]));
}
// dense[key[i]] = val[i] for i < *count (keys unique)
__global__ __launch_bounds__(kBlock) void k_scatter_dense(const uint64_t *__restrict__ key, const Fr *__restrict__ val,
                                                          const unsigned int *__restrict__ count, Fr *__restrict__ dense) {
    const uint64_t n = *count;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) fr_st(dense + key[i], fr_ld(val + i));
}

// The two initialisations without the full sorts (used by sc_gkr_prove, which needs neither f1(g,.,.) as a list nor any order):
//   phase one  a_hg[x]   = sum over non-zeros (z,x,y) of eq(g,z) * v * f3[y]          (mod.rs:30-38, with f1_g expanded)
//   phase two  f1_gu[y]  = sum over non-zeros (z,x,y) of eq(g,z) * eq(u,x) * v         (mod.rs:62 + to_dense)
// Field addition is exact, so the sums may be taken in any order and grouping.  The non-zeros are bucketed by the high bits of the
// target cell (ONE radix pass over those bits; none for phase two when the list arrives sorted by index, the order a BTreeMap gives);
// one workgroup per bucket of 2^c cells keeps the cells in LDS as eight uint64 lanes of 32-bit limbs each (the wide format of
// the sharded path), adds every term with LDS integer atomics and reduces each cell mod p once at the end.  Global 64-bit atomics
// on the dense table were measured first: 8.4 M of them take 0.4 ms on this part (profiles/r2d_gkr_init.txt), slower than sorting.
__device__ __forceinline__ Fr wide_fold_cell(const uint64_t lane[8]) { // V = sum_j lane_j 2^(32 j) (lanes < 2^63) -> V mod p
    Fr lo;
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t t = lane[j] + carry; // < 2^63 + 2^32: no wrap
        lo.v[j] = (uint32_t)t;
        carry = t >> 32;
    }
    lo = scd::fr_reduce_once(scd::fr_reduce_once(lo)); // lo < 2^256 < 3p
    // V = lo + carry * 2^256 and carry * 2^256 mod p = mont_mul(carry, R^2)
    Fr hi = scd::fr_zero(), r2;
    hi.v[0] = (uint32_t)carry;
    hi.v[1] = (uint32_t)(carry >> 32);
    const uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r2.v[2 * q] = (uint32_t)R2[q];
        r2.v[2 * q + 1] = (uint32_t)(R2[q] >> 32);
    }
    return scd::fr_add(lo, scd::fr_mul(hi, r2));
}
// eq(point, b) as the product of two small tables (k_eq_halves): b's low kl bits and the rest.  2 x 2^11 entries at most stay in
// L2, where a 2^dim-entry table is a random 32-byte read from memory per non-zero.
struct EqSplit {
    const Fr *lo, *hi;
    uint32_t kl;
};
__device__ __forceinline__ Fr eq_at(const EqSplit &e, const uint64_t b) {
    const Fr l = fr_ld(e.lo + (b & ((1ULL << e.kl) - 1)));
    return e.hi ? scd::fr_mul(fr_ld(e.hi + (b >> e.kl)), l) : l;
}
// kPhase 1: index (z,x,y), cell = x, term = eq(g,z) * v * f3[y].  kPhase 2: index (z,x,y), cell = y, term = eq(g,z) * eq(u,x) * v.
// kPhase 3 (initialize_phase_two on the list f1(g,.,.)): index (x,y), cell = y, term = eq(u,x) * v.
// kPhase 4 (phase two of sc_gkr_prove after a bucketed phase one): as 2, with v already multiplied by eq(g,z) -- phase one's scatter pass
// computed eq(g,z) * v for its own term and left it behind (the reference's f1(g,.,.), one entry per non-zero, unmerged): term = eq(u,x) * a.
template <int kPhase>
__host__ __device__ constexpr uint32_t cell_shift_of(const uint32_t dim) { return (kPhase == 2 || kPhase == 4) ? 2 * dim : dim; }
template <int kPhase>
__device__ __forceinline__ Fr gkr_term(const uint64_t id, const Fr &v, const uint32_t dim, const EqSplit &eg, const EqSplit &eu, const Fr *__restrict__ f3) {
    const uint64_t mask = (1ULL << dim) - 1;
    if (kPhase == 3) return scd::fr_mul(eq_at(eu, id & mask), v);
    if (kPhase == 4) return scd::fr_mul(eq_at(eu, (id >> dim) & mask), v);
    const Fr a = scd::fr_mul(eq_at(eg, id & mask), v);
    if (kPhase == 1) return scd::fr_mul(a, fr_ld(f3 + (id >> (2 * dim))));
    return scd::fr_mul(a, eq_at(eu, (id >> dim) & mask));
}

// Grouping the terms by bucket (= target cell >> c): a counting sort over the bucket bits, radix-sort style and without global
// atomics.  Block b owns the contiguous chunk [b * chunk, (b + 1) * chunk) of the list in both sweeps.
//   k_bucket_count    LDS histogram of the chunk -> counts[b][bucket]
//   k_bucket_colscan  one thread per bucket: counts[.][bucket] -> exclusive prefix over the blocks, total[bucket]
//   k_bucket_scan     one block: start[bucket] = exclusive prefix of the totals, start[nb] = n
//   k_bucket_scatter  recomputes each non-zero's term and writes (term, cell within the bucket) behind start + prefix; the order
//                     inside a bucket is whatever the LDS rank atomics give (the sums do not care)
// (rocprim's radix_sort_pairs restricted to the bucket bits runs as a 21-launch merge sort at this size, 0.18 ms.)
constexpr int kMaxBuckets = 2048, kSortBlock = 1024;
__device__ __forceinline__ uint32_t bucket_of(const uint64_t id, const uint32_t shift, const uint32_t nb_mask) { return (uint32_t)(id >> shift) & nb_mask; }
__global__ __launch_bounds__(kSortBlock) void k_bucket_count(const uint64_t *__restrict__ idx, const uint64_t n, const uint64_t chunk, const uint32_t shift,
                                                             const uint32_t nb_mask, uint32_t *__restrict__ counts) {
    __shared__ uint32_t hist[kMaxBuckets];
    for (uint32_t b = threadIdx.x; b <= nb_mask; b += kSortBlock) hist[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += kSortBlock) atomicAdd(&hist[bucket_of(idx[i], shift, nb_mask)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b <= nb_mask; b += kSortBlock) counts[(uint64_t)blockIdx.x * (nb_mask + 1) + b] = hist[b];
}
__global__ __launch_bounds__(kBlock) void k_bucket_colscan(uint32_t *__restrict__ counts, const uint32_t n_blocks, const uint32_t nb, uint64_t *__restrict__ total) {
    const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= nb) return;
    uint64_t run = 0;
    constexpr int kRows = 32; // rows in flight (the loop is load latency)
    for (uint32_t r0 = 0; r0 < n_blocks; r0 += kRows) {
        uint32_t c[kRows];
#pragma unroll
        for (int q = 0; q < kRows; ++q) c[q] = r0 + q < n_blocks ? counts[(uint64_t)(r0 + q) * nb + b] : 0;
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            if (r0 + q < n_blocks) counts[(uint64_t)(r0 + q) * nb + b] = (uint32_t)run; // (the list holds < 2^32 non-zeros)
            run += c[q];
        }
    }
    total[b] = run;
}
// one block of kMaxBuckets / 2 threads: start[b] = sum of total[0 .. b), start[nb] = n
__global__ __launch_bounds__(kMaxBuckets / 2) void k_bucket_scan(const uint64_t *__restrict__ total, const uint32_t nb, uint64_t *__restrict__ start) {
    __shared__ uint64_t part[kMaxBuckets / 2];
    const uint32_t t = threadIdx.x;
    const uint64_t a = 2 * t < nb ? total[2 * t] : 0, b = 2 * t + 1 < nb ? total[2 * t + 1] : 0;
    part[t] = a + b;
    __syncthreads();
    for (uint32_t off = 1; off < kMaxBuckets / 2; off <<= 1) { // inclusive scan of the pair sums
        const uint64_t add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    const uint64_t before = part[t] - (a + b);
    if (2 * t < nb) start[2 * t] = before;
    if (2 * t + 1 < nb) start[2 * t + 1] = before + a;
    if (t == kMaxBuckets / 2 - 1) start[nb] = part[t];
}
template <int kPhase>
__global__ __launch_bounds__(kSortBlock) void k_bucket_scatter(const uint64_t *__restrict__ idx, const Fr *__restrict__ vals, const uint64_t n, const uint64_t chunk,
                                                               const uint32_t dim, const uint32_t c, const EqSplit eg, const EqSplit eu,
                                                               const Fr *__restrict__ f3, const uint32_t *__restrict__ counts,
                                                               const uint64_t *__restrict__ start, Fr *__restrict__ out_term,
                                                               uint16_t *__restrict__ out_cell, Fr *__restrict__ out_a) {
    __shared__ uint32_t rank[kMaxBuckets];
    __shared__ uint64_t base[kMaxBuckets];
    const uint32_t cell_shift = cell_shift_of<kPhase>(dim), nb_mask = (1u << (dim - c)) - 1;
    for (uint32_t b = threadIdx.x; b <= nb_mask; b += kSortBlock) {
        rank[b] = 0;
        base[b] = start[b] + counts[(uint64_t)blockIdx.x * (nb_mask + 1) + b];
    }
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
    constexpr int kIlp = 4; // four independent product chains per lane: a term is three or four dependent Montgomery products
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (uint64_t)kIlp * kSortBlock) {
        uint64_t id[kIlp];
        Fr t[kIlp];
#pragma unroll
        for (int q = 0; q < kIlp; ++q) {
            const uint64_t i = min(i0 + (uint64_t)q * kSortBlock, hi - 1); // (clamped: the surplus lanes recompute the last entry and drop it)
            id[q] = idx[i];
            t[q] = fr_ld(vals + i);
        }
        if (kPhase == 1 && out_a) { // eq(g,z) * v is phase two's input: kept, in list order (the clamped surplus lanes rewrite the last entry with its own value)
#pragma unroll
            for (int q = 0; q < kIlp; ++q) {
                t[q] = scd::fr_mul(eq_at(eg, id[q] & ((1ULL << dim) - 1)), t[q]);
                fr_st(out_a + min(i0 + (uint64_t)q * kSortBlock, hi - 1), t[q]);
            }
#pragma unroll
            for (int q = 0; q < kIlp; ++q) t[q] = scd::fr_mul(t[q], fr_ld(f3 + (id[q] >> (2 * dim))));
        } else {
#pragma unroll
            for (int q = 0; q < kIlp; ++q) t[q] = gkr_term<kPhase>(id[q], t[q], dim, eg, eu, f3);
        }
#pragma unroll
        for (int q = 0; q < kIlp; ++q) {
            if (i0 + (uint64_t)q * kSortBlock >= hi) continue;
            const uint32_t b = bucket_of(id[q], cell_shift + c, nb_mask);
            const uint64_t pos = base[b] + atomicAdd(&rank[b], 1u);
            fr_st(out_term + pos, t[q]);
            out_cell[pos] = (uint16_t)((id[q] >> cell_shift) & ((1u << c) - 1));
        }
    }
}
// start offsets of a list that is already grouped (in non-decreasing bucket order): position i opens every bucket in
// (bucket(i - 1), bucket(i)]; i = n closes the rest
__global__ __launch_bounds__(kBlock) void k_bucket_bounds(const uint64_t *__restrict__ idx, const uint64_t n, const uint32_t shift, const uint32_t nb_mask,
                                                          uint64_t *__restrict__ start) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i <= n; i += stride) {
        const int64_t prev = i == 0 ? -1 : (int64_t)bucket_of(idx[i - 1], shift, nb_mask);
        const int64_t cur = i == n ? (int64_t)nb_mask + 1 : (int64_t)bucket_of(idx[i], shift, nb_mask);
        for (int64_t b = prev + 1; b <= cur; ++b) start[b] = i;
    }
}
// One workgroup per bucket, bucket b at positions [start[b], start[b + 1]): the 2^c cells live in LDS as eight uint64 lanes each.
// terms != null: (term, cell) pairs written by k_bucket_scatter; terms == null: the list itself is grouped (phase two on an
// index-ordered list) and the terms are computed here.  A bucket with more than max_entries non-zeros raises *skewed and is left
// alone (the caller falls back to the list form).
template <int kPhase>
__global__ __launch_bounds__(kBlock) void k_bucket_accumulate(const Fr *__restrict__ terms, const uint16_t *__restrict__ cells, const uint64_t *__restrict__ idx,
                                                              const Fr *__restrict__ vals, const uint32_t dim, const uint32_t c, const EqSplit eg,
                                                              const EqSplit eu, const Fr *__restrict__ f3, const uint64_t *__restrict__ start,
                                                              const uint64_t max_entries, unsigned int *__restrict__ skewed, Fr *__restrict__ dense) {
    extern __shared__ uint64_t cell[]; // lane-major: cell[j << c | i] (neighbouring cells in neighbouring banks)
    const uint64_t lo = start[blockIdx.x], hi = start[blockIdx.x + 1];
    for (uint32_t i = threadIdx.x; i < (8u << c); i += kBlock) cell[i] = 0;
    __syncthreads();
    if (hi - lo > max_entries) {
        if (threadIdx.x == 0) atomicOr(skewed, 1u);
        return;
    }
    const uint32_t cell_shift = cell_shift_of<kPhase>(dim);
    if (terms) {
        for (uint64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
            const Fr t = fr_ld(terms + i);
            const uint32_t ci = cells[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(reinterpret_cast<unsigned long long *>(cell + (((uint32_t)j << c) | ci)), (unsigned long long)t.v[j]);
        }
    } else {
        constexpr int kIlp = 2; // two independent product chains per lane (a bucket holds about two non-zeros per lane)
        for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (uint64_t)kIlp * kBlock) {
            uint64_t id[kIlp];
            Fr t[kIlp];
#pragma unroll
            for (int q = 0; q < kIlp; ++q) {
                const uint64_t i = min(i0 + (uint64_t)q * kBlock, hi - 1);
                id[q] = idx[i];
                t[q] = fr_ld(vals + i);
            }
#pragma unroll
            for (int q = 0; q < kIlp; ++q) t[q] = gkr_term<kPhase>(id[q], t[q], dim, eg, eu, f3);
#pragma unroll
            for (int q = 0; q < kIlp; ++q) {
                if (i0 + (uint64_t)q * kBlock >= hi) continue;
                const uint32_t ci = (uint32_t)((id[q] >> cell_shift) & ((1u << c) - 1));
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(reinterpret_cast<unsigned long long *>(cell + (((uint32_t)j << c) | ci)), (unsigned long long)t[q].v[j]);
            }
        }
    }
    __syncthreads();
    for (uint32_t ci = threadIdx.x; ci < (1u << c); ci += kBlock) {
        uint64_t lane[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) lane[j] = cell[((uint32_t)j << c) | ci];
        fr_st(dense + (((uint64_t)blockIdx.x << c) | ci), wide_fold_cell(lane));
    }
}

// out[i] = in[i] * (*scalar) (mod.rs:71-75 with f2(u) still on the device)
__global__ __launch_bounds__(kBlock) void k_scale_by(const Fr *__restrict__ in, const Fr *__restrict__ scalar, const uint64_t n, Fr *__restrict__ out) {
    const Fr sc = fr_ld(scalar);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) fr_st(out + i, scd::fr_mul(fr_ld(in + i), sc));
}

// flag |= 1 if any idx[i] has a bit at or above `bits` (index range check of device-resident inputs); |= 2 if the list is not in
// non-decreasing index order
__global__ __launch_bounds__(kBlock) void k_idx_range(const uint64_t *__restrict__ idx, const uint64_t n, const uint32_t bits, unsigned int *__restrict__ flag) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    bool bad = false, unsorted = false;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t id = idx[i];
        bad |= (id >> bits) != 0;
        unsorted |= i > 0 && idx[i - 1] > id;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
    if (__any(unsorted) && (threadIdx.x & 63) == 0) atomicOr(flag, 2u);
}

// canonical elements -> 8 zero-extended 32-bit limbs in uint64 lanes (summable across ranks with an integer all-reduce) ...
__global__ __launch_bounds__(kBlock) void k_widen(const Fr *__restrict__ in, const uint64_t n, uint64_t *__restrict__ lanes) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const Fr a = fr_ld(in + i);
        uint4 *o = reinterpret_cast<uint4 *>(lanes + 8 * i);
        o[0] = make_uint4(a.v[0], 0u, a.v[1], 0u);
        o[1] = make_uint4(a.v[2], 0u, a.v[3], 0u);
        o[2] = make_uint4(a.v[4], 0u, a.v[5], 0u);
        o[3] = make_uint4(a.v[6], 0u, a.v[7], 0u);
    }
}
// ... and back: V = sum_j lane_j 2^(32 j) (lanes < 2^63) = lo + hi 2^256, V mod p = (lo mod p) + hi * (2^256 mod p).  The device
// twin of sc_wide_reduce (abi.hip).
__global__ __launch_bounds__(kBlock) void k_wide_fold(const uint64_t *__restrict__ lanes, const uint64_t n, Fr *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint64_t lane[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) lane[j] = lanes[8 * i + j];
        fr_st(out + i, wide_fold_cell(lane));
    }
}

#define G_TRY(expr)                                                                                                      \
    do {                                                                                                                 \
        hipError_t e_ = (expr);                                                                                          \
        if (e_ != hipSuccess) {                                                                                          \
            (void)hipGetLastError(); /* (not left behind as the thread's sticky error: see prover_internal.hpp, HIP_TRY) */          \
            return sc_internal_fail(e_ == hipErrorOutOfMemory ? SC_ERR_OOM : SC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                    hipGetErrorString(e_), __FILE__, __LINE__);                                          \
        }                                                                                                                \
    } while (0)

struct DevBuf { // device scratch for the lifetime of one API call: one arena (hipFree is synchronous and slow: ~20 separate
                // buffers cost more than the kernels), bump-allocated, with a plain hipMalloc fallback if the estimate is short
    char *arena = nullptr;
    size_t cap = 0, used = 0;
    bool leased = false; // the arena belongs to the process-wide cache (GkrCache below): not freed here
    std::vector<void *> extra;
    ~DevBuf() {
        if (arena && !leased) (void)hipFree(arena);
        for (void *p : extra) (void)hipFree(p);
        release_lease();
    }
    void release_lease();
    hipError_t reserve(size_t bytes); // takes the cached arena when it is free and large enough (or grows it), else allocates
    template <typename T>
    hipError_t alloc(T **out, size_t n) {
        const size_t bytes = ((n ? n : 1) * sizeof(T) + 255) & ~size_t(255);
        if (used + bytes <= cap) {
            *out = reinterpret_cast<T *>(arena + used);
            used += bytes;
            return hipSuccess;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) {
            extra.push_back(p);
            *out = static_cast<T *>(p);
        }
        return e;
    }
};
// Process-wide scratch cache: a GKR call needs ~1 GB of device scratch for dim = 20 and a two-table prover handle; allocating
// and freeing them costs more than a millisecond per call (hipMalloc + the synchronous hipFree), a fifth of config 5.  One
// call at a time may hold the cache (try_lock; concurrent callers simply allocate their own).  sc_release_caches() frees it.
struct GkrCache {
    std::mutex mu;
    char *arena = nullptr;
    size_t cap = 0;
    int device = -1;
    sc_prover *prover = nullptr; // K = 1, M = 2, U = 2 borrowing handle of `prover_dim` variables
    uint32_t prover_dim = 0;
    hipStream_t side = nullptr;  // sc_gkr_prove: phase two's bucket plan (indices only) runs beside phase one
    unsigned int *h_pin = nullptr; // ... and leaves its verdict in this pinned word
};
static GkrCache g_cache;
static thread_local bool t_holds_cache = false;

hipError_t DevBuf::reserve(size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (bytes <= sc_internal_cache_limit() && !t_holds_cache && g_cache.mu.try_lock()) { // (over sc_set_cache_limit: this call's own arena, freed at its end)
        t_holds_cache = true;
        leased = true;
        if (g_cache.device != dev || g_cache.cap < bytes) {
            if (g_cache.prover) { // the cached handle borrows tables inside the old arena and lives on the old device
                sc_prover_free(g_cache.prover);
                g_cache.prover = nullptr;
            }
            if (g_cache.arena) (void)hipFree(g_cache.arena);
            if (g_cache.side && g_cache.device != dev) {
                (void)hipStreamDestroy(g_cache.side);
                g_cache.side = nullptr;
            }
            g_cache.arena = nullptr;
            g_cache.cap = 0;
            g_cache.device = dev;
            if (hipMalloc(reinterpret_cast<void **>(&g_cache.arena), bytes) == hipSuccess) g_cache.cap = bytes;
            else (void)hipGetLastError();
        }
        arena = g_cache.arena;
        cap = g_cache.cap;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&arena), bytes);
    if (e == hipSuccess) cap = bytes;
    else (void)hipGetLastError();
    return hipSuccess; // on failure fall back to per-buffer allocations
}
void DevBuf::release_lease() {
    if (leased) {
        leased = false;
        t_holds_cache = false;
        g_cache.mu.unlock();
    }
}

extern "C" int sc_release_caches(void) {
    sc_internal_release_eval_cache();
    sc_internal_release_handle_pool();
    std::lock_guard<std::mutex> lk(g_cache.mu);
    if (g_cache.device >= 0) (void)hipSetDevice(g_cache.device);
    if (g_cache.prover) sc_prover_free(g_cache.prover);
    g_cache.prover = nullptr;
    if (g_cache.arena) (void)hipFree(g_cache.arena);
    if (g_cache.side) (void)hipStreamDestroy(g_cache.side);
    if (g_cache.h_pin) (void)hipHostFree(g_cache.h_pin);
    g_cache.h_pin = nullptr;
    g_cache.side = nullptr;
    g_cache.arena = nullptr;
    g_cache.cap = 0;
    g_cache.device = -1;
    return SC_OK;
}

inline size_t gkr_scratch_estimate(uint64_t nnz, uint64_t N) { return (size_t)(704 + 64) * (nnz + 1) + (size_t)(352 + 64) * N + ((size_t)64 << 20); }

inline int grid_for(uint64_t n) { return scd::grid_for_pairs(n); }
inline FrHost hostfr(const sch::Fr &a) {
    FrHost h;
    std::memcpy(&h, &a, 32);
    return h;
}

// Sort (idx, vals) by idx on the device.  out_idx / out_vals are n-element device buffers.
int sort_sparse(DevBuf &mem, const uint64_t *d_idx, const Fr *d_vals, uint64_t n, uint32_t bits, uint64_t *out_idx, Fr *out_vals,
                hipStream_t s) {
    if (n == 0) return SC_OK;
    uint32_t *iota = nullptr, *perm = nullptr;
    G_TRY(mem.alloc(&iota, n));
    G_TRY(mem.alloc(&perm, n));
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kBlock), 0, s, iota, n);
    size_t tmp_bytes = 0;
    G_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_idx, out_idx, iota, perm, n, 0, bits, s));
    char *tmp = nullptr;
    G_TRY(mem.alloc(&tmp, tmp_bytes));
    G_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, d_idx, out_idx, iota, perm, n, 0, bits, s));
    hipLaunchKernelGGL(k_gather, dim3(grid_for(n)), dim3(kBlock), 0, s, d_vals, perm, n, out_vals);
    G_TRY(hipGetLastError());
    return SC_OK;
}

// eq(point, b) for all b in {0,1}^k (1 <= k <= kEqMaxVars) -> eq[0 .. 2^k)
int build_eq_table(DevBuf &mem, const sch::Fr *point, uint32_t k, Fr *eq, hipStream_t s) {
    EqPoint P;
    std::memset(&P, 0, sizeof(P));
    for (uint32_t i = 0; i < k; ++i) P.g[i] = hostfr(point[i]);
    const int kl = (int)std::min<uint32_t>(k, kEqHalfMax), kh = (int)k - kl;
    if (kh == 0) {
        hipLaunchKernelGGL(k_eq_halves, dim3(kEqSlices, 1), dim3(kEqBlock), 0, s, eq, (Fr *)nullptr, P, kl, 0);
    } else {
        Fr *lo = nullptr, *hi = nullptr;
        G_TRY(mem.alloc(&lo, (size_t)1 << kl));
        G_TRY(mem.alloc(&hi, (size_t)1 << kh));
        hipLaunchKernelGGL(k_eq_halves, dim3(kEqSlices, 2), dim3(kEqBlock), 0, s, lo, hi, P, kl, kh);
        hipLaunchKernelGGL(k_eq_outer, dim3(grid_for(1ULL << k)), dim3(kBlock), 0, s, lo, hi, kl, 1ULL << k, eq);
    }
    G_TRY(hipGetLastError());
    return SC_OK;
}

// SparseMultilinearExtension::fix_variables over the low k variables of a list sorted by index.
// Output: merged (key, value) list (sorted, unique keys) in out_key/out_val (capacity n) and *d_count on device.
int sparse_fix(DevBuf &mem, const uint64_t *d_idx, const Fr *d_vals, uint64_t n, const sch::Fr *point, uint32_t k, uint64_t *out_key,
               Fr *out_val, unsigned int *d_count, hipStream_t s) {
    if (n == 0) {
        G_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned int), s));
        return SC_OK;
    }
    uint64_t *key = nullptr;
    Fr *w = nullptr;
    G_TRY(mem.alloc(&key, n));
    G_TRY(mem.alloc(&w, n));
    const bool use_table = k >= 1 && k <= (uint32_t)kEqMaxVars && (1ULL << k) <= 8 * n + 1024;
    if (k == 0) {
        G_TRY(hipMemcpyAsync(key, d_idx, n * 8, hipMemcpyDeviceToDevice, s));
        G_TRY(hipMemcpyAsync(w, d_vals, n * 32, hipMemcpyDeviceToDevice, s));
    } else if (use_table) {
        Fr *eq = nullptr;
        G_TRY(mem.alloc(&eq, (size_t)1 << k));
        int rc_eq = build_eq_table(mem, point, k, eq, s);
        if (rc_eq) return rc_eq;
        hipLaunchKernelGGL(k_sparse_scale, dim3(grid_for(n)), dim3(kBlock), 0, s, d_idx, d_vals, (const uint32_t *)nullptr, eq, k, n, key, w);
    } else {
        Fr *d_point = nullptr;
        G_TRY(mem.alloc(&d_point, k));
        G_TRY(hipMemcpyAsync(d_point, point, (size_t)k * 32, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_sparse_scale_direct, dim3(grid_for(n)), dim3(kBlock), 0, s, d_idx, d_vals, (const uint32_t *)nullptr, d_point, k, n,
                           key, w);
    }
    G_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    G_TRY(rocprim::reduce_by_key(nullptr, tmp_bytes, key, w, n, out_key, out_val, d_count, FrAdd(), rocprim::equal_to<uint64_t>(), s));
    char *tmp = nullptr;
    G_TRY(mem.alloc(&tmp, tmp_bytes));
    G_TRY(rocprim::reduce_by_key(tmp, tmp_bytes, key, w, n, out_key, out_val, d_count, FrAdd(), rocprim::equal_to<uint64_t>(), s));
    return SC_OK;
}

// initialize_phase_one on the device (inputs sorted by index).  Outputs: d_hg (2^dim), f1_g list + count.
int phase_one_device(DevBuf &mem, const uint64_t *d_idx, const Fr *d_vals, uint64_t nnz, uint32_t dim, const Fr *d_f3, const sch::Fr *g,
                     Fr *d_hg, uint64_t *d_f1g_idx, Fr *d_f1g_vals, unsigned int *d_n1, uint64_t *h_n1, hipStream_t s) {
    scd::plan_hit(scd::kPlanGkrListForm);
    int rc = sparse_fix(mem, d_idx, d_vals, nnz, g, dim, d_f1g_idx, d_f1g_vals, d_n1, s); // mod.rs:31
    if (rc) return rc;
    unsigned int n1 = 0;
    G_TRY(hipMemcpyAsync(&n1, d_n1, sizeof(n1), hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
    *h_n1 = n1;
    const uint64_t N = 1ULL << dim;
    G_TRY(hipMemsetAsync(d_hg, 0, N * 32, s)); // a_hg starts at zero (mod.rs:30)
    if (n1 == 0) return SC_OK;
    uint64_t *key_x = nullptr, *key_xs = nullptr, *ux = nullptr;
    Fr *t = nullptr, *ts = nullptr, *sums = nullptr;
    uint32_t *iota = nullptr, *perm = nullptr;
    unsigned int *d_cnt = nullptr;
    G_TRY(mem.alloc(&key_x, n1));
    G_TRY(mem.alloc(&key_xs, n1));
    G_TRY(mem.alloc(&ux, n1));
    G_TRY(mem.alloc(&t, n1));
    G_TRY(mem.alloc(&ts, n1));
    G_TRY(mem.alloc(&sums, n1));
    G_TRY(mem.alloc(&iota, n1));
    G_TRY(mem.alloc(&perm, n1));
    G_TRY(mem.alloc(&d_cnt, 1));
    hipLaunchKernelGGL(k_hg_terms, dim3(grid_for(n1)), dim3(kBlock), 0, s, d_f1g_idx, d_f1g_vals, d_f3, dim, (uint64_t)n1, key_x, t); // mod.rs:34-36
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n1)), dim3(kBlock), 0, s, iota, (uint64_t)n1);
    size_t tb = 0;
    G_TRY(rocprim::radix_sort_pairs(nullptr, tb, key_x, key_xs, iota, perm, n1, 0, dim ? dim : 1, s));
    char *tmp = nullptr;
    G_TRY(mem.alloc(&tmp, tb));
    G_TRY(rocprim::radix_sort_pairs(tmp, tb, key_x, key_xs, iota, perm, n1, 0, dim ? dim : 1, s));
    hipLaunchKernelGGL(k_gather, dim3(grid_for(n1)), dim3(kBlock), 0, s, t, perm, (uint64_t)n1, ts);
    size_t tb2 = 0;
    G_TRY(rocprim::reduce_by_key(nullptr, tb2, key_xs, ts, n1, ux, sums, d_cnt, FrAdd(), rocprim::equal_to<uint64_t>(), s));
    char *tmp2 = nullptr;
    G_TRY(mem.alloc(&tmp2, tb2));
    G_TRY(rocprim::reduce_by_key(tmp2, tb2, key_xs, ts, n1, ux, sums, d_cnt, FrAdd(), rocprim::equal_to<uint64_t>(), s));
    hipLaunchKernelGGL(k_scatter_dense, dim3(grid_for(n1)), dim3(kBlock), 0, s, ux, sums, d_cnt, d_hg);
    G_TRY(hipGetLastError());
    return SC_OK;
}

// initialize_phase_two on the device: dense 2^dim table of f1(g,u,.)
int phase_two_device(DevBuf &mem, const uint64_t *d_f1g_idx, const Fr *d_f1g_vals, uint64_t n1, uint32_t dim, const sch::Fr *u, Fr *d_out,
                     hipStream_t s) {
    scd::plan_hit(scd::kPlanGkrListForm);
    const uint64_t N = 1ULL << dim;
    G_TRY(hipMemsetAsync(d_out, 0, N * 32, s));
    if (n1 == 0) return SC_OK;
    uint64_t *ky = nullptr;
    Fr *vy = nullptr;
    unsigned int *d_cnt = nullptr;
    G_TRY(mem.alloc(&ky, n1));
    G_TRY(mem.alloc(&vy, n1));
    G_TRY(mem.alloc(&d_cnt, 1));
    int rc = sparse_fix(mem, d_f1g_idx, d_f1g_vals, n1, u, dim, ky, vy, d_cnt, s); // mod.rs:62
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_dense, dim3(grid_for(n1)), dim3(kBlock), 0, s, ky, vy, d_cnt, d_out); // to_dense_multilinear_extension
    G_TRY(hipGetLastError());
    return SC_OK;
}

int check_gkr_args(uint64_t nnz, uint32_t dim) {
    if (dim == 0) return sc_internal_fail(SC_ERR_CONSTANT_POLY, "Attempt to prove a constant.");
    if (dim > 21) return sc_internal_fail(SC_ERR_BAD_ARG, "dim %u: 3*dim index bits do not fit 64-bit indices", dim);
    if (nnz >= (1ULL << 32)) return sc_internal_fail(SC_ERR_BAD_ARG, "nnz too large");
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    G_TRY(hipSetDevice(sc_internal_device())); // scratch, kernels and the prover handle of this call all live on the thread's device
    return SC_OK;
}
int check_points(const uint64_t *pt, uint32_t n, const char *what) {
    for (uint32_t i = 0; i < n; ++i) {
        sch::Fr a;
        std::memcpy(&a, pt + 4 * i, 32);
        if (sch::geq_p(a)) return sc_internal_fail(SC_ERR_BAD_ARG, "%s[%u] is not a canonical field element", what, i);
    }
    return SC_OK;
}

// inputs: host arrays are staged into the call's scratch, device arrays (flags & SC_TABLES_ON_DEVICE) are used where they are
template <typename T>
int stage_in(DevBuf &mem, const void *src, uint64_t n, bool on_device, const T **out, hipStream_t s) {
    if (on_device || n == 0) {
        *out = static_cast<const T *>(src);
        return SC_OK;
    }
    T *d = nullptr;
    G_TRY(mem.alloc(&d, n));
    G_TRY(hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    *out = d;
    return SC_OK;
}
// every index below 2^bits (bits < 64); host arrays are checked on the host, device arrays by a kernel
// (meanwhile: work the caller wants on the device while the host waits for the verdict; it must not depend on the indices being valid)
int check_index_range(DevBuf &mem, const uint64_t *idx, uint64_t n, uint32_t bits, bool on_device, const char *what, hipStream_t s,
                      bool *sorted_out = nullptr, const std::function<int()> *meanwhile = nullptr) {
    if (sorted_out) *sorted_out = true;
    if (bits >= 64 || n == 0 || !on_device) {
        int rc_m = meanwhile ? (*meanwhile)() : SC_OK;
        if (rc_m) return rc_m;
    }
    if (bits >= 64 || n == 0) return SC_OK;
    if (!on_device) {
        bool sorted = true;
        for (uint64_t i = 0; i < n; ++i) {
            if ((idx[i] >> bits) != 0) return sc_internal_fail(SC_ERR_BAD_ARG, "%s index %llu out of range", what, (unsigned long long)i);
            sorted &= i == 0 || idx[i - 1] <= idx[i];
        }
        if (sorted_out) *sorted_out = sorted;
        return SC_OK;
    }
    unsigned int *d_flag = nullptr, h = 0;
    G_TRY(mem.alloc(&d_flag, 1));
    G_TRY(hipMemsetAsync(d_flag, 0, sizeof(unsigned int), s));
    hipLaunchKernelGGL(k_idx_range, dim3(grid_for(n)), dim3(kBlock), 0, s, idx, n, bits, d_flag);
    // (into pinned memory where the caller holds the cache's page: a copy into pageable memory waits for the device inside the call)
    unsigned int *h_dst = (mem.leased && g_cache.h_pin) ? g_cache.h_pin + 1 : &h;
    G_TRY(hipMemcpyAsync(h_dst, d_flag, sizeof(h), hipMemcpyDeviceToHost, s));
    if (meanwhile) {
        int rc_m = (*meanwhile)();
        if (rc_m) return rc_m;
    }
    G_TRY(hipStreamSynchronize(s));
    h = *h_dst;
    if (h & 1u) return sc_internal_fail(SC_ERR_BAD_ARG, "%s has an index out of range", what);
    if (sorted_out) *sorted_out = !(h & 2u);
    return SC_OK;
}

// eq(point, .) over dim variables as two small tables (EqSplit) in `mem`
int build_eq_split(DevBuf &mem, const sch::Fr *point, uint32_t dim, EqSplit *out, hipStream_t s) {
    EqPoint P;
    std::memset(&P, 0, sizeof(P));
    for (uint32_t i = 0; i < dim; ++i) P.g[i] = hostfr(point[i]);
    const int kl = (int)std::min<uint32_t>(dim, kEqHalfMax), kh = (int)dim - kl;
    Fr *lo = nullptr, *hi = nullptr;
    G_TRY(mem.alloc(&lo, (size_t)1 << kl));
    if (kh > 0) G_TRY(mem.alloc(&hi, (size_t)1 << kh));
    hipLaunchKernelGGL(k_eq_halves, dim3(kEqSlices, kh > 0 ? 2 : 1), dim3(kEqBlock), 0, s, lo, hi, P, kl, kh);
    G_TRY(hipGetLastError());
    out->lo = lo;
    out->hi = hi;
    out->kl = (uint32_t)kl;
    return SC_OK;
}

// 2^dim cells are worth a pass of their own when the list is not much sparser than the table (and the eq tables fit EqPoint).
// sc_set_policy("gkr_direct", 0) switches the bucketed form off (tests of the list form).
bool bucketed_form_pays(uint64_t nnz, uint32_t dim) {
    return scd::policy(scd::kPolGkrDirect) != 0 && nnz > 0 && dim <= (uint32_t)kEqMaxVars && (1ULL << dim) <= 8 * nnz + 1024;
}

// One of the two dense tables of sc_gkr_prove through the bucketed kernels, in two steps.
//   bucket_plan  what depends on the INDICES only: the bucket offsets (a grouped list: k_bucket_bounds; else the counting sort's histogram,
//                column scan and prefix) and whether any bucket is too crowded for one workgroup (k_bucket_skew -> *h_skew once `st` has
//                been synchronised).  Phase two's plan needs no challenge: sc_gkr_prove runs it beside phase one, on its second stream.
//   bucket_fill  what depends on the points: the terms (k_bucket_scatter for an ungrouped list) and the cells' sums (k_bucket_accumulate):
//                one or two launches, no host synchronisation -- the caller chains whatever comes next on the same stream.
// A skewed plan (a pathological index distribution) leaves `dense` unfinished: the caller takes the list form instead.
struct BucketPlan {
    uint32_t c = 0, n_blocks = 0;
    uint64_t nb = 0, chunk = 0, max_entries = 0;
    uint64_t *start = nullptr;
    uint32_t *counts = nullptr;
    Fr *terms = nullptr;
    uint16_t *cells = nullptr;
    unsigned int *d_skew = nullptr;
    unsigned int *h_skew = nullptr; // the caller's word (pinned memory if the copy must not block the host); valid once the plan's stream has been synchronised
    bool grouped = false, ok = false; // ok = false: outside the bucketed form (the list form takes the list)
};
__global__ __launch_bounds__(kBlock) void k_bucket_skew(const uint64_t *__restrict__ start, const uint64_t nb, const uint64_t max_entries, unsigned int *__restrict__ skewed) {
    const uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b < nb && start[b + 1] - start[b] > max_entries) atomicOr(skewed, 1u);
}
template <int kPhase>
int bucket_plan(DevBuf &mem, const uint64_t *d_idx, uint64_t n, uint32_t dim, bool idx_sorted, BucketPlan *P, unsigned int *h_skew, hipStream_t st) {
    P->h_skew = h_skew;
    *h_skew = 0;
    // 2^c cells per bucket, at most kMaxBuckets buckets: dim 20 -> 2048 buckets of 512 cells (32 KB of lanes, four workgroups per CU)
    P->c = (uint32_t)std::max<int>((int)dim - 11, 0);
    if (P->c > 10) return sc_internal_fail(SC_ERR_BAD_ARG, "dim %u is outside the bucketed form", dim);
    P->ok = n < (1ULL << 32); // (positions and per-block counts are 32-bit here: the list form takes a longer list)
    if (!P->ok) return SC_OK;
    P->nb = 1ULL << (dim - P->c);
    P->max_entries = std::max<uint64_t>(8192, 64 * (n / P->nb + 1));
    const uint32_t shift = cell_shift_of<kPhase>(dim) + P->c, nb_mask = (uint32_t)(P->nb - 1);
    G_TRY(mem.alloc(&P->start, P->nb + 1));
    G_TRY(mem.alloc(&P->d_skew, 1));
    G_TRY(hipMemsetAsync(P->d_skew, 0, sizeof(unsigned int), st));
    P->grouped = kPhase != 1 && idx_sorted; // index order is y-major: already grouped by phase two's cells
    scd::plan_hit(P->grouped ? scd::kPlanGkrBucketedGrouped : scd::kPlanGkrBucketedCounted);
    if (P->grouped) {
        hipLaunchKernelGGL(k_bucket_bounds, dim3(grid_for(n + 1)), dim3(kBlock), 0, st, d_idx, n, shift, nb_mask, P->start);
    } else {
        P->n_blocks = (uint32_t)std::min<uint64_t>(256, (n + 4095) / 4096); // (one 1024-thread block per CU at 2^20 non-zeros)
        P->chunk = (n + P->n_blocks - 1) / P->n_blocks;
        uint64_t *total = nullptr;
        G_TRY(mem.alloc(&P->counts, (size_t)P->n_blocks * P->nb));
        G_TRY(mem.alloc(&total, P->nb));
        G_TRY(mem.alloc(&P->terms, n));
        G_TRY(mem.alloc(&P->cells, n));
        hipLaunchKernelGGL(k_bucket_count, dim3(P->n_blocks), dim3(kSortBlock), 0, st, d_idx, n, P->chunk, shift, nb_mask, P->counts);
        hipLaunchKernelGGL(k_bucket_colscan, dim3((unsigned)((P->nb + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, P->counts, P->n_blocks, (uint32_t)P->nb, total);
        hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(kMaxBuckets / 2), 0, st, total, (uint32_t)P->nb, P->start);
    }
    hipLaunchKernelGGL(k_bucket_skew, dim3((unsigned)((P->nb + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, P->start, P->nb, P->max_entries, P->d_skew);
    G_TRY(hipGetLastError());
    G_TRY(hipMemcpyAsync(P->h_skew, P->d_skew, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    return SC_OK;
}
template <int kPhase>
int bucket_fill(const BucketPlan &P, const uint64_t *d_idx, const Fr *d_vals, uint64_t n, uint32_t dim, const EqSplit &eg, const EqSplit &eu, const Fr *d_f3,
                Fr *dense, hipStream_t st, Fr *out_a = nullptr) {
    if (!P.grouped)
        hipLaunchKernelGGL((k_bucket_scatter<kPhase>), dim3(P.n_blocks), dim3(kSortBlock), 0, st, d_idx, d_vals, n, P.chunk, dim, P.c, eg, eu, d_f3, P.counts, P.start,
                           P.terms, P.cells, out_a);
    const size_t lds = ((size_t)8 << P.c) * sizeof(uint64_t);
    hipLaunchKernelGGL((k_bucket_accumulate<kPhase>), dim3((unsigned)P.nb), dim3(kBlock), lds, st, P.terms, P.cells, d_idx, d_vals, dim, P.c, eg, eu, d_f3, P.start,
                       P.max_entries, P.d_skew, dense);
    G_TRY(hipGetLastError());
    return SC_OK;
}
// plan + fill + one synchronisation (the stand-alone initialisations, and phase one of sc_gkr_prove).  *done = false: the list form must take over.
template <int kPhase>
int bucketed_dense(DevBuf &mem, const uint64_t *d_idx, const Fr *d_vals, uint64_t n, uint32_t dim, const EqSplit &eg, const EqSplit &eu, const Fr *d_f3,
                   bool idx_sorted, Fr *dense, bool *done, hipStream_t s, Fr *out_a = nullptr) {
    BucketPlan P;
    unsigned int h_skew = 0;
    int rc = bucket_plan<kPhase>(mem, d_idx, n, dim, idx_sorted, &P, &h_skew, s);
    if (rc == SC_OK && P.ok) rc = bucket_fill<kPhase>(P, d_idx, d_vals, n, dim, eg, eu, d_f3, dense, s, out_a);
    G_TRY(hipStreamSynchronize(s)); // (also on the error path: the copy into h_skew must not outlive it)
    if (rc) return rc;
    *done = P.ok && h_skew == 0;
    return SC_OK;
}

} // namespace

extern "C" int sc_gkr_phase_one(const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz, uint32_t dim, const uint64_t *f3,
                                const uint64_t *g, uint32_t flags, uint64_t *h_g, uint64_t *f1g_idx, uint64_t *f1g_vals, uint64_t *f1g_nnz) {
    if ((nnz && (!f1_idx || !f1_vals)) || !f3 || !g || !h_g || !f1g_nnz || (nnz && (!f1g_idx || !f1g_vals)))
        return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    GkrGate gate_;
    int rc = check_gkr_args(nnz, dim);
    if (rc) return rc;
    if ((rc = check_points(g, dim, "g"))) return rc;
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    hipStream_t s = nullptr;
    DevBuf mem;
    const uint64_t N = 1ULL << dim;
    (void)mem.reserve(gkr_scratch_estimate(nnz, N));
    bool f1_sorted = true;
    if ((rc = check_index_range(mem, f1_idx, nnz, 3 * dim, dev, "f1", s, &f1_sorted))) return rc;
    const uint64_t *d_idx = nullptr;
    const Fr *d_vals = nullptr, *d_f3 = nullptr;
    uint64_t *d_idx_s = nullptr, *d_gi = nullptr;
    Fr *d_vals_s = nullptr, *d_hg = nullptr, *d_gv = nullptr;
    unsigned int *d_n1 = nullptr;
    if ((rc = stage_in(mem, f1_idx, nnz, dev, &d_idx, s)) || (rc = stage_in(mem, f1_vals, nnz, dev, &d_vals, s)) || (rc = stage_in(mem, f3, N, dev, &d_f3, s)))
        return rc;
    G_TRY(mem.alloc(&d_idx_s, nnz));
    G_TRY(mem.alloc(&d_vals_s, nnz));
    G_TRY(mem.alloc(&d_n1, 1));
    if (dev) { // results are produced in place
        d_hg = reinterpret_cast<Fr *>(h_g);
        d_gi = f1g_idx;
        d_gv = reinterpret_cast<Fr *>(f1g_vals);
    } else {
        G_TRY(mem.alloc(&d_hg, N));
        G_TRY(mem.alloc(&d_gi, nnz));
        G_TRY(mem.alloc(&d_gv, nnz));
    }
    // f1(g,.,.) comes back as a merged list, which needs the non-zeros in index order (a list that arrives ordered, as a BTreeMap's does,
    // is not sorted again); a_hg does not: it is bucketed from the list as it came whenever that pays (sc_gkr_prove's initialisation)
    const uint64_t *si = d_idx;
    const Fr *sv = d_vals;
    if (!f1_sorted) {
        if ((rc = sort_sparse(mem, d_idx, d_vals, nnz, 3 * dim, d_idx_s, d_vals_s, s))) return rc;
        si = d_idx_s;
        sv = d_vals_s;
    }
    uint64_t n1 = 0;
    bool hg_done = false;
    if (bucketed_form_pays(nnz, dim)) {
        EqSplit eq_g{nullptr, nullptr, 0}, none{nullptr, nullptr, 0};
        if ((rc = build_eq_split(mem, reinterpret_cast<const sch::Fr *>(g), dim, &eq_g, s))) return rc;
        if ((rc = bucketed_dense<1>(mem, d_idx, d_vals, nnz, dim, eq_g, none, d_f3, f1_sorted, d_hg, &hg_done, s))) return rc; // mod.rs:30-38
    }
    if (hg_done) {
        if ((rc = sparse_fix(mem, si, sv, nnz, reinterpret_cast<const sch::Fr *>(g), dim, d_gi, d_gv, d_n1, s))) return rc; // mod.rs:31
        unsigned int h_n1 = 0;
        G_TRY(hipMemcpyAsync(&h_n1, d_n1, sizeof(h_n1), hipMemcpyDeviceToHost, s));
        G_TRY(hipStreamSynchronize(s));
        n1 = h_n1;
    } else if ((rc = phase_one_device(mem, si, sv, nnz, dim, d_f3, reinterpret_cast<const sch::Fr *>(g), d_hg, d_gi, d_gv, d_n1, &n1, s))) {
        return rc;
    }
    if (!dev) {
        G_TRY(hipMemcpyAsync(h_g, d_hg, N * 32, hipMemcpyDeviceToHost, s));
        if (n1) {
            G_TRY(hipMemcpyAsync(f1g_idx, d_gi, n1 * 8, hipMemcpyDeviceToHost, s));
            G_TRY(hipMemcpyAsync(f1g_vals, d_gv, n1 * 32, hipMemcpyDeviceToHost, s));
        }
    }
    G_TRY(hipStreamSynchronize(s));
    *f1g_nnz = n1;
    return SC_OK;
}

extern "C" int sc_gkr_phase_two(const uint64_t *f1g_idx, const uint64_t *f1g_vals, uint64_t nnz, uint32_t dim, const uint64_t *u,
                                uint32_t flags, uint64_t *f1_gu) {
    if ((nnz && (!f1g_idx || !f1g_vals)) || !u || !f1_gu) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    GkrGate gate_;
    int rc = check_gkr_args(nnz, dim);
    if (rc) return rc;
    if ((rc = check_points(u, dim, "u"))) return rc;
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    hipStream_t s = nullptr;
    DevBuf mem;
    const uint64_t N = 1ULL << dim;
    (void)mem.reserve(gkr_scratch_estimate(nnz, N));
    bool f1g_sorted = true;
    if ((rc = check_index_range(mem, f1g_idx, nnz, 2 * dim, dev, "f1_g", s, &f1g_sorted))) return rc;
    const uint64_t *d_idx = nullptr;
    const Fr *d_vals = nullptr;
    uint64_t *d_idx_s = nullptr;
    Fr *d_vals_s = nullptr, *d_out = nullptr;
    if ((rc = stage_in(mem, f1g_idx, nnz, dev, &d_idx, s)) || (rc = stage_in(mem, f1g_vals, nnz, dev, &d_vals, s))) return rc;
    G_TRY(mem.alloc(&d_idx_s, nnz));
    G_TRY(mem.alloc(&d_vals_s, nnz));
    if (dev) d_out = reinterpret_cast<Fr *>(f1_gu);
    else G_TRY(mem.alloc(&d_out, N));
    bool done = false;
    if (bucketed_form_pays(nnz, dim)) { // the dense table straight from the list, in whatever order it came (ordered: no grouping pass)
        EqSplit eq_u{nullptr, nullptr, 0}, none{nullptr, nullptr, 0};
        if ((rc = build_eq_split(mem, reinterpret_cast<const sch::Fr *>(u), dim, &eq_u, s))) return rc;
        if ((rc = bucketed_dense<3>(mem, d_idx, d_vals, nnz, dim, none, eq_u, nullptr, f1g_sorted, d_out, &done, s))) return rc; // mod.rs:62
    }
    if (!done) {
        if ((rc = sort_sparse(mem, d_idx, d_vals, nnz, 2 * dim, d_idx_s, d_vals_s, s))) return rc;
        if ((rc = phase_two_device(mem, d_idx_s, d_vals_s, nnz, dim, reinterpret_cast<const sch::Fr *>(u), d_out, s))) return rc;
    }
    if (!dev) G_TRY(hipMemcpyAsync(f1_gu, d_out, N * 32, hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

int sc_internal_allreduce_lanes(sc_comm *c, uint64_t *d_lanes, size_t n_words, hipStream_t s); // comm.hip
int sc_internal_comm_ranks(sc_comm *c);

// ---- f4: GKR initialisation with f1's non-zeros spread over several GPUs -------------------------------------------
// Every rank holds a disjoint subset of f1's (index, value) pairs -- any partition -- and all of f3.  a_hg[x] = sum over ALL
// non-zeros, so a rank's scatter is a partial sum: the dense tables of the ranks are added with one table-sized integer
// all-reduce of the widened limbs (64 bytes per entry; no modular all-reduce exists) and folded back mod p on the device.
// f1(g,.,.) stays distributed: each rank keeps the fold of ITS entries (keys may repeat across ranks; the true value of a key is
// the sum over ranks), and initialize_phase_two folds and scatters it locally again before the same all-reduce.
// `lanes_or_null`: when given, the rank's contribution is left there as lanes (2^dim x 8 uint64, host or device per flags) and no
// communication happens (the caller all-reduces: sumcheck_amd/sharded.py over torch.distributed); otherwise `comm` does it.
static int gkr_dense_allreduce(DevBuf &mem, sc_comm *comm, Fr *d_table, uint64_t N, uint64_t *lanes_or_null, bool dev, hipStream_t s) {
    if (!lanes_or_null && sc_internal_comm_ranks(comm) == 1) return SC_OK;
    uint64_t *d_lanes = nullptr;
    if (lanes_or_null && dev) d_lanes = lanes_or_null;
    else G_TRY(mem.alloc(&d_lanes, 8 * N));
    hipLaunchKernelGGL(k_widen, dim3(grid_for(N)), dim3(kBlock), 0, s, d_table, N, d_lanes);
    G_TRY(hipGetLastError());
    if (lanes_or_null) {
        if (!dev) G_TRY(hipMemcpyAsync(lanes_or_null, d_lanes, N * 64, hipMemcpyDeviceToHost, s));
        G_TRY(hipStreamSynchronize(s));
        return SC_OK;
    }
    int rc = sc_internal_allreduce_lanes(comm, d_lanes, 8 * N, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_wide_fold, dim3(grid_for(N)), dim3(kBlock), 0, s, d_lanes, N, d_table);
    G_TRY(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_wide_reduce_table(const uint64_t *lanes, uint64_t n, uint64_t *out, uint32_t flags) {
    if (n && (!lanes || !out)) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (n == 0) return SC_OK;
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    GkrGate gate_;
    G_TRY(hipSetDevice(sc_internal_device()));
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    hipStream_t s = nullptr;
    DevBuf mem;
    const uint64_t *d_l = lanes;
    Fr *d_o = reinterpret_cast<Fr *>(out);
    if (!dev) {
        uint64_t *tmp = nullptr;
        G_TRY(mem.alloc(&tmp, 8 * n));
        G_TRY(mem.alloc(&d_o, n));
        G_TRY(hipMemcpyAsync(tmp, lanes, n * 64, hipMemcpyHostToDevice, s));
        d_l = tmp;
    }
    hipLaunchKernelGGL(k_wide_fold, dim3(grid_for(n)), dim3(kBlock), 0, s, d_l, n, d_o);
    G_TRY(hipGetLastError());
    if (!dev) G_TRY(hipMemcpyAsync(out, d_o, n * 32, hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

extern "C" int sc_gkr_phase_one_sharded(sc_comm *comm_or_null, const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz_local, uint32_t dim,
                                        const uint64_t *f3, const uint64_t *g, uint32_t flags, uint64_t *h_g_or_null, uint64_t *lanes_or_null,
                                        uint64_t *f1g_idx, uint64_t *f1g_vals, uint64_t *f1g_nnz) {
    if ((nnz_local && (!f1_idx || !f1_vals)) || !f3 || !g || !f1g_nnz || (nnz_local && (!f1g_idx || !f1g_vals)) || (!h_g_or_null && !lanes_or_null))
        return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    GkrGate gate_;
    int rc = check_gkr_args(nnz_local, dim);
    if (rc) return rc;
    if ((rc = check_points(g, dim, "g"))) return rc;
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    hipStream_t s = nullptr;
    DevBuf mem;
    const uint64_t N = 1ULL << dim;
    (void)mem.reserve(gkr_scratch_estimate(nnz_local, N) + 64 * N);
    if ((rc = check_index_range(mem, f1_idx, nnz_local, 3 * dim, dev, "f1", s))) return rc;
    const uint64_t *d_idx = nullptr;
    const Fr *d_vals = nullptr, *d_f3 = nullptr;
    uint64_t *d_idx_s = nullptr, *d_gi = nullptr;
    Fr *d_vals_s = nullptr, *d_hg = nullptr, *d_gv = nullptr;
    unsigned int *d_n1 = nullptr;
    if ((rc = stage_in(mem, f1_idx, nnz_local, dev, &d_idx, s)) || (rc = stage_in(mem, f1_vals, nnz_local, dev, &d_vals, s)) ||
        (rc = stage_in(mem, f3, N, dev, &d_f3, s)))
        return rc;
    G_TRY(mem.alloc(&d_idx_s, nnz_local));
    G_TRY(mem.alloc(&d_vals_s, nnz_local));
    G_TRY(mem.alloc(&d_n1, 1));
    const bool hg_in_place = dev && h_g_or_null;
    if (hg_in_place) d_hg = reinterpret_cast<Fr *>(h_g_or_null);
    else G_TRY(mem.alloc(&d_hg, N));
    if (dev) {
        d_gi = f1g_idx;
        d_gv = reinterpret_cast<Fr *>(f1g_vals);
    } else {
        G_TRY(mem.alloc(&d_gi, nnz_local));
        G_TRY(mem.alloc(&d_gv, nnz_local));
    }
    if ((rc = sort_sparse(mem, d_idx, d_vals, nnz_local, 3 * dim, d_idx_s, d_vals_s, s))) return rc;
    uint64_t n1 = 0;
    if ((rc = phase_one_device(mem, d_idx_s, d_vals_s, nnz_local, dim, d_f3, reinterpret_cast<const sch::Fr *>(g), d_hg, d_gi, d_gv, d_n1, &n1, s))) return rc;
    if ((rc = gkr_dense_allreduce(mem, comm_or_null, d_hg, N, lanes_or_null, dev, s))) return rc;
    if (h_g_or_null && !hg_in_place) G_TRY(hipMemcpyAsync(h_g_or_null, d_hg, N * 32, hipMemcpyDeviceToHost, s));
    if (!dev && n1) {
        G_TRY(hipMemcpyAsync(f1g_idx, d_gi, n1 * 8, hipMemcpyDeviceToHost, s));
        G_TRY(hipMemcpyAsync(f1g_vals, d_gv, n1 * 32, hipMemcpyDeviceToHost, s));
    }
    G_TRY(hipStreamSynchronize(s));
    *f1g_nnz = n1;
    return SC_OK;
}

extern "C" int sc_gkr_phase_two_sharded(sc_comm *comm_or_null, const uint64_t *f1g_idx, const uint64_t *f1g_vals, uint64_t nnz_local, uint32_t dim,
                                        const uint64_t *u, uint32_t flags, uint64_t *f1_gu_or_null, uint64_t *lanes_or_null) {
    if ((nnz_local && (!f1g_idx || !f1g_vals)) || !u || (!f1_gu_or_null && !lanes_or_null)) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    GkrGate gate_;
    int rc = check_gkr_args(nnz_local, dim);
    if (rc) return rc;
    if ((rc = check_points(u, dim, "u"))) return rc;
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    hipStream_t s = nullptr;
    DevBuf mem;
    const uint64_t N = 1ULL << dim;
    (void)mem.reserve(gkr_scratch_estimate(nnz_local, N) + 64 * N);
    if ((rc = check_index_range(mem, f1g_idx, nnz_local, 2 * dim, dev, "f1_g", s))) return rc;
    const uint64_t *d_idx = nullptr;
    const Fr *d_vals = nullptr;
    uint64_t *d_idx_s = nullptr;
    Fr *d_vals_s = nullptr, *d_out = nullptr;
    if ((rc = stage_in(mem, f1g_idx, nnz_local, dev, &d_idx, s)) || (rc = stage_in(mem, f1g_vals, nnz_local, dev, &d_vals, s))) return rc;
    G_TRY(mem.alloc(&d_idx_s, nnz_local));
    G_TRY(mem.alloc(&d_vals_s, nnz_local));
    const bool in_place = dev && f1_gu_or_null;
    if (in_place) d_out = reinterpret_cast<Fr *>(f1_gu_or_null);
    else G_TRY(mem.alloc(&d_out, N));
    if ((rc = sort_sparse(mem, d_idx, d_vals, nnz_local, 2 * dim, d_idx_s, d_vals_s, s))) return rc;
    if ((rc = phase_two_device(mem, d_idx_s, d_vals_s, nnz_local, dim, reinterpret_cast<const sch::Fr *>(u), d_out, s))) return rc;
    if ((rc = gkr_dense_allreduce(mem, comm_or_null, d_out, N, lanes_or_null, dev, s))) return rc;
    if (f1_gu_or_null && !in_place) G_TRY(hipMemcpyAsync(f1_gu_or_null, d_out, N * 32, hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

extern "C" int sc_dense_scale(const uint64_t *in, uint64_t n, const uint64_t *scalar, uint64_t *out, uint32_t flags) {
    if ((n && (!in || !out)) || !scalar) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    int rc = check_points(scalar, 1, "scalar");
    if (rc) return rc;
    if (n == 0) return SC_OK;
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    GkrGate gate_;
    G_TRY(hipSetDevice(sc_internal_device()));
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    hipStream_t s = nullptr;
    sch::Fr sv;
    std::memcpy(&sv, scalar, 32);
    if (dev) {
        G_TRY(scd::launch_scale(reinterpret_cast<const uint4 *>(in), reinterpret_cast<uint4 *>(out), hostfr(sv), n, s));
        G_TRY(hipStreamSynchronize(s));
        return SC_OK;
    }
    DevBuf mem;
    Fr *d_in = nullptr, *d_out = nullptr;
    G_TRY(mem.alloc(&d_in, n));
    G_TRY(mem.alloc(&d_out, n));
    G_TRY(hipMemcpyAsync(d_in, in, n * 32, hipMemcpyHostToDevice, s));
    G_TRY(scd::launch_scale(reinterpret_cast<const uint4 *>(d_in), reinterpret_cast<uint4 *>(d_out), hostfr(sv), n, s));
    G_TRY(hipMemcpyAsync(out, d_out, n * 32, hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

// SparseMultilinearExtension::evaluate(point) = fix_variables(point)[0] (ark-poly; used by GKRRoundSumcheckSubClaim::
// verify_subclaim, src/gkr_round_sumcheck/data_structures.rs:33-56): every non-zero is weighted by eq(point, index) on the device
// (num_vars products per entry) and the weighted values are summed with the segmented field sum over the single key 0.
extern "C" int sc_sparse_evaluate(const uint64_t *idx, const uint64_t *vals, uint64_t nnz, uint32_t num_vars, const uint64_t *point,
                                  uint64_t *out) {
    if ((nnz && (!idx || !vals)) || (num_vars && !point) || !out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (num_vars > 63) return sc_internal_fail(SC_ERR_BAD_ARG, "num_vars %u: indices are 64-bit", num_vars);
    if (nnz >= (1ULL << 32)) return sc_internal_fail(SC_ERR_BAD_ARG, "nnz too large");
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    GkrGate gate_;
    G_TRY(hipSetDevice(sc_internal_device()));
    int rc;
    if ((rc = check_points(point, num_vars, "point"))) return rc;
    for (uint64_t i = 0; i < nnz; ++i)
        if ((idx[i] >> num_vars) != 0) return sc_internal_fail(SC_ERR_BAD_ARG, "index %llu out of range", (unsigned long long)i);
    std::memset(out, 0, 32);
    if (nnz == 0) return SC_OK;
    hipStream_t s = nullptr;
    DevBuf mem;
    (void)mem.reserve((size_t)160 * (nnz + 1) + ((size_t)16 << 20));
    uint64_t *d_idx = nullptr, *d_key = nullptr, *d_okey = nullptr;
    Fr *d_vals = nullptr, *d_w = nullptr, *d_oval = nullptr, *d_point = nullptr;
    unsigned int *d_cnt = nullptr;
    G_TRY(mem.alloc(&d_idx, nnz));
    G_TRY(mem.alloc(&d_key, nnz));
    G_TRY(mem.alloc(&d_okey, nnz));
    G_TRY(mem.alloc(&d_vals, nnz));
    G_TRY(mem.alloc(&d_w, nnz));
    G_TRY(mem.alloc(&d_oval, nnz));
    G_TRY(mem.alloc(&d_point, num_vars));
    G_TRY(mem.alloc(&d_cnt, 1));
    G_TRY(hipMemcpyAsync(d_idx, idx, nnz * 8, hipMemcpyHostToDevice, s));
    G_TRY(hipMemcpyAsync(d_vals, vals, nnz * 32, hipMemcpyHostToDevice, s));
    if (num_vars) G_TRY(hipMemcpyAsync(d_point, point, (size_t)num_vars * 32, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sparse_scale_direct, dim3(grid_for(nnz)), dim3(kBlock), 0, s, d_idx, d_vals, (const uint32_t *)nullptr, d_point, num_vars, nnz,
                       d_key, d_w);
    G_TRY(hipGetLastError());
    size_t tb = 0;
    G_TRY(rocprim::reduce_by_key(nullptr, tb, d_key, d_w, nnz, d_okey, d_oval, d_cnt, FrAdd(), rocprim::equal_to<uint64_t>(), s));
    char *tmp = nullptr;
    G_TRY(mem.alloc(&tmp, tb));
    G_TRY(rocprim::reduce_by_key(tmp, tb, d_key, d_w, nnz, d_okey, d_oval, d_cnt, FrAdd(), rocprim::equal_to<uint64_t>(), s));
    G_TRY(hipMemcpyAsync(out, d_oval, 32, hipMemcpyDeviceToHost, s)); // every key is 0: one segment
    G_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

// One sumcheck phase: product 1*(A*B) over two device tables (start_phase{1,2}_sumcheck, mod.rs:45-54,66-82), dim rounds
// of prove_round / feed / sample (mod.rs:111-119,126-133).  The handle (stream, ping-pong buffers, pinned result page) is
// created for phase one and rewound onto phase two's tables, so the second phase allocates nothing.
static int run_phase(sch::Blake2b512Rng &rng, sc_prover **handle, const Fr *dA, const Fr *dB, uint32_t dim, uint64_t *out_msgs,
                     sch::Fr *challenges, const std::function<int(sc_prover *)> *after_reset = nullptr) {
    const uint64_t *tabs[2] = {reinterpret_cast<const uint64_t *>(dA), reinterpret_cast<const uint64_t *>(dB)};
    int rc;
    if (*handle == nullptr) {
        const uint32_t offs[2] = {0, 2}, idx[2] = {0, 1};
        sc_poly_desc d;
        std::memset(&d, 0, sizeof(d));
        d.num_vars = dim;
        d.max_multiplicands = 2;
        d.n_products = 1;
        d.coeffs = sch::kOne.l; // F::one()
        d.prod_offsets = offs;
        d.prod_indices = idx;
        d.n_tables = 2;
        d.tables = tabs;
        d.flags = SC_TABLES_ON_DEVICE | SC_TABLES_BORROW; // the inputs are this call's own scratch: no second copy
        rc = sc_prover_init(&d, handle);
    } else {
        rc = sc_prover_reset(*handle, tabs, SC_TABLES_ON_DEVICE);
    }
    if (rc) return rc;
    if (after_reset && (rc = (*after_reset)(*handle))) return rc; // (phase two: the coefficient f2(u), on the handle's stream)
    return sc_internal_run_rounds(*handle, rng, dim, out_msgs, challenges); // prove_round / feed / sample x dim (late rounds pipelined)
}

// GKRRoundSumcheck::prove (mod.rs:93-139) with f1's non-zeros spread over the ranks of `comm` AND both sumcheck phases sharded: the two
// initialisations as sc_gkr_phase_one/two_sharded (every rank ends with the complete dense table), then each phase's product of two
// tables proved like any sharded MLSumcheck -- rank r takes entries [r 2^dim / G, (r + 1) 2^dim / G) of both tables (high-bit sharding:
// pairs stay local under LSB-first binding), per-round all-reduce of the three evaluations, early gather, replicated tail.  f2(u) is
// evaluated on every rank (2^dim products; the scalar is the same everywhere) and scales the rank's own slice of f3.  Worth it from
// dim ~ 24; BASELINE config 5 (dim = 20) stays on one GPU.  Every rank returns the same proof and (u, v).
int sc_internal_sharded_phase(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, uint32_t nv_total, uint64_t *out_proof, uint64_t *out_randomness); // api.hip
int sc_internal_comm_rank(sc_comm *c);
extern "C" int sc_gkr_prove_sharded(sc_comm *comm, sc_rng *rng, const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz_local, uint32_t dim,
                                    const uint64_t *f2, const uint64_t *f3, const uint64_t *g, uint32_t flags, uint64_t *out_proof, uint64_t *out_uv_or_null) {
    if (!comm || !rng || (nnz_local && (!f1_idx || !f1_vals)) || !f2 || !f3 || !g || !out_proof) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    int rc = check_gkr_args(nnz_local, dim);
    if (rc) return rc;
    if ((rc = check_points(g, dim, "g"))) return rc;
    const uint32_t G = (uint32_t)sc_internal_comm_ranks(comm), rank = (uint32_t)sc_internal_comm_rank(comm);
    scd::plan_hit(scd::kPlanGkrSharded);
    uint32_t k = 0;
    while ((1u << k) < G) ++k;
    if ((1u << k) != G || dim <= k) return sc_internal_fail(SC_ERR_BAD_ARG, "the number of ranks must be a power of two below 2^dim");
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    const uint64_t N = 1ULL << dim, Nl = N >> k;
    G_TRY(hipSetDevice(sc_internal_device()));
    struct Bufs { // plain allocations: the calls below lease the process-wide GKR scratch themselves
        std::vector<void *> v;
        ~Bufs() {
            for (void *q : v) (void)hipFree(q);
        }
        hipError_t get(void **out, size_t bytes) {
            hipError_t e = hipMalloc(out, bytes ? bytes : 32);
            if (e == hipSuccess) v.push_back(*out);
            return e;
        }
    } bufs;
    const uint64_t *d_idx = f1_idx, *d_vals = f1_vals, *d_f2 = f2, *d_f3 = f3;
    auto stage = [&](const uint64_t *src, size_t bytes, const uint64_t **dst) -> int {
        void *d = nullptr;
        G_TRY(bufs.get(&d, bytes));
        if (bytes) G_TRY(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        *dst = static_cast<const uint64_t *>(d);
        return SC_OK;
    };
    if (!dev) {
        if ((rc = stage(f1_idx, nnz_local * 8, &d_idx)) || (rc = stage(f1_vals, nnz_local * 32, &d_vals)) || (rc = stage(f2, N * 32, &d_f2)) ||
            (rc = stage(f3, N * 32, &d_f3)))
            return rc;
    }
    uint64_t *d_hg = nullptr, *d_gi = nullptr, *d_gv = nullptr, *d_f1gu = nullptr, *d_f3s = nullptr, *d_scalar = nullptr;
    G_TRY(bufs.get(reinterpret_cast<void **>(&d_hg), N * 32));
    G_TRY(bufs.get(reinterpret_cast<void **>(&d_gi), nnz_local * 8));
    G_TRY(bufs.get(reinterpret_cast<void **>(&d_gv), nnz_local * 32));
    G_TRY(bufs.get(reinterpret_cast<void **>(&d_f1gu), N * 32));
    G_TRY(bufs.get(reinterpret_cast<void **>(&d_f3s), Nl * 32));
    G_TRY(bufs.get(reinterpret_cast<void **>(&d_scalar), 32));
    uint64_t n1 = 0;
    if ((rc = sc_gkr_phase_one_sharded(comm, d_idx, d_vals, nnz_local, dim, d_f3, g, SC_TABLES_ON_DEVICE, d_hg, nullptr, d_gi, d_gv, &n1))) return rc; // mod.rs:106
    // phase one: h_g * f2 (mod.rs:45-54, 107-119) on this rank's slices
    struct Handle {
        sc_prover *p = nullptr;
        ~Handle() {
            if (p) sc_prover_free(p);
        }
    } h;
    const uint64_t *tabs[2] = {d_hg + 4 * (size_t)rank * Nl, d_f2 + 4 * (size_t)rank * Nl};
    {
        const uint32_t offs[2] = {0, 2}, idx[2] = {0, 1};
        sc_poly_desc d;
        std::memset(&d, 0, sizeof(d));
        d.num_vars = dim - k;
        d.max_multiplicands = 2;
        d.n_products = 1;
        d.coeffs = sch::kOne.l;
        d.prod_offsets = offs;
        d.prod_indices = idx;
        d.n_tables = 2;
        d.tables = tabs;
        d.flags = SC_TABLES_ON_DEVICE | SC_TABLES_BORROW;
        if ((rc = sc_prover_init(&d, &h.p))) return rc;
    }
    std::vector<uint64_t> u((size_t)dim * 4), v((size_t)dim * 4);
    if ((rc = sc_internal_sharded_phase(h.p, comm, rng->rng, dim, out_proof, u.data()))) return rc;
    // phase two: f1(g, u, .) * (f2(u) f3) (mod.rs:57-82, 121-133)
    if ((rc = sc_gkr_phase_two_sharded(comm, d_gi, d_gv, n1, dim, u.data(), SC_TABLES_ON_DEVICE, d_f1gu, nullptr))) return rc;
    if ((rc = sc_fix_variables(d_f2, dim, u.data(), dim, d_scalar, SC_TABLES_ON_DEVICE))) return rc; // f2.evaluate(&u), mod.rs:122
    uint64_t scalar[4];
    G_TRY(hipMemcpy(scalar, d_scalar, 32, hipMemcpyDeviceToHost));
    if ((rc = sc_dense_scale(d_f3 + 4 * (size_t)rank * Nl, Nl, scalar, d_f3s, SC_TABLES_ON_DEVICE))) return rc; // mod.rs:71-75, this rank's slice
    const uint64_t *tabs2[2] = {d_f1gu + 4 * (size_t)rank * Nl, d_f3s};
    if ((rc = sc_prover_reset(h.p, tabs2, SC_TABLES_ON_DEVICE))) return rc;
    if ((rc = sc_internal_sharded_phase(h.p, comm, rng->rng, dim, out_proof + (size_t)dim * 12, v.data()))) return rc;
    if (out_uv_or_null) {
        std::memcpy(out_uv_or_null, u.data(), (size_t)dim * 32);
        std::memcpy(out_uv_or_null + (size_t)dim * 4, v.data(), (size_t)dim * 32);
    }
    return SC_OK;
}

namespace {
struct ProverGuard { // declared AFTER the DevBuf it pairs with, so it is destroyed first, while the cache lease is still held
    sc_prover *p = nullptr;
    uint32_t dim = 0;
    bool cacheable = false;
    void take_cached(uint32_t d, bool leased) {
        dim = d;
        cacheable = leased;
        if (leased && g_cache.prover && g_cache.prover_dim == d) {
            p = g_cache.prover;
            g_cache.prover = nullptr;
        }
    }
    ~ProverGuard() {
        if (!p) return;
        if (cacheable && t_holds_cache) {
            if (g_cache.prover) sc_prover_free(g_cache.prover);
            g_cache.prover = p;
            g_cache.prover_dim = dim;
        } else {
            sc_prover_free(p);
        }
    }
};
} // namespace

extern "C" int sc_gkr_prove(sc_rng *rng, const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz, uint32_t dim, const uint64_t *f2,
                            const uint64_t *f3, const uint64_t *g, uint32_t flags, uint64_t *out_proof, uint64_t *out_uv_or_null) {
    if (!rng || (nnz && (!f1_idx || !f1_vals)) || !f2 || !f3 || !g || !out_proof) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    GkrGate gate_;
    int rc = check_gkr_args(nnz, dim);
    if (rc) return rc;
    if ((rc = check_points(g, dim, "g"))) return rc;
    const bool dev = flags & SC_TABLES_ON_DEVICE;
    hipStream_t s = nullptr;
    DevBuf mem;
    (void)mem.reserve(gkr_scratch_estimate(nnz, 1ULL << dim));
    bool f1_sorted = true;
    if (mem.leased) { // the cache's second stream and pinned page (the index check's verdict and phase two's plan verdict land there)
        if (!g_cache.side && hipStreamCreateWithFlags(&g_cache.side, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            g_cache.side = nullptr;
        }
        if (g_cache.side && !g_cache.h_pin && hipHostMalloc(reinterpret_cast<void **>(&g_cache.h_pin), 64, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            g_cache.h_pin = nullptr;
        }
    }
    // the eq(g, .) tables need only g: they are built while the host waits for the verdict on the indices
    bool direct = bucketed_form_pays(nnz, dim);
    EqSplit eq_g{nullptr, nullptr, 0}, eq_u{nullptr, nullptr, 0};
    const std::function<int()> eq_meanwhile = [&]() -> int { return direct ? build_eq_split(mem, reinterpret_cast<const sch::Fr *>(g), dim, &eq_g, s) : SC_OK; };
    if ((rc = check_index_range(mem, f1_idx, nnz, 3 * dim, dev, "f1", s, &f1_sorted, &eq_meanwhile))) return rc;
    const bool trace = std::getenv("SC_GKR_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gkr] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const uint64_t N = 1ULL << dim;
    const uint64_t *d_idx = nullptr;
    const Fr *d_vals = nullptr, *d_f2 = nullptr, *d_f3 = nullptr;
    uint64_t *d_idx_s = nullptr, *d_gi = nullptr;
    Fr *d_vals_s = nullptr, *d_hg = nullptr, *d_gv = nullptr, *d_f1gu = nullptr, *d_a = nullptr;
    unsigned int *d_n1 = nullptr;
    G_TRY(mem.alloc(&d_idx_s, nnz));
    G_TRY(mem.alloc(&d_vals_s, nnz));
    G_TRY(mem.alloc(&d_hg, N));
    G_TRY(mem.alloc(&d_gi, nnz));
    G_TRY(mem.alloc(&d_gv, nnz));
    G_TRY(mem.alloc(&d_f1gu, N));
    G_TRY(mem.alloc(&d_n1, 1));
    lap("alloc");
    if ((rc = stage_in(mem, f1_idx, nnz, dev, &d_idx, s)) || (rc = stage_in(mem, f1_vals, nnz, dev, &d_vals, s)) ||
        (rc = stage_in(mem, f2, N, dev, &d_f2, s)) || (rc = stage_in(mem, f3, N, dev, &d_f3, s)))
        return rc;
    // Host inputs: the copies above are asynchronous on `s`, and the second stream (non-blocking: no implicit ordering with `s`) reads the
    // staged indices below while phase one runs -- it must not start before they have landed.  (Found in round 4 as a one-in-fifteen
    // mismatch of phase two's messages at dim = 1 in the GPU suite; device-resident inputs were never affected.)
    if (!dev) G_TRY(hipStreamSynchronize(s));
    lap("h2d");
    // Bucketed initialisation (k_bucket_accumulate) whenever the list is dense enough for 2^dim cells to be worth a pass; the list form (sort, merge, scatter:
    // what sc_gkr_phase_one returns to a caller) otherwise, and when a bucket is too crowded.  sc_set_policy("gkr_direct", 0) forces the list form.
    hipStream_t side = s;
    unsigned int skew2_local = 0, *h_skew2 = &skew2_local;
    if (mem.leased) {
        if (g_cache.side && g_cache.h_pin) { // (a copy into pinned memory does not hold the host: the plan below runs beside phase one)
            side = g_cache.side;
            h_skew2 = g_cache.h_pin;
        }
    }
    BucketPlan plan2; // (declared before `join`: the second stream's copy into *h_skew2 is over before either goes away)
    struct StreamJoin { // no return path leaves work behind on the second stream (it reads and writes this call's scratch)
        hipStream_t st;
        ~StreamJoin() {
            if (st) (void)hipStreamSynchronize(st);
        }
    } join{side != s ? side : nullptr};
    uint64_t n1 = 0;
    bool have_list = false; // f1(g,.,.) as a merged list in d_gi / d_gv (list form only)
    // Phase two's bucket plan depends on the indices only: it runs now, on the second stream, while phase one has the device (a call without
    // the cache's stream runs it in line; phase one's synchronisation below covers it then).
    if (direct && (rc = bucket_plan<2>(mem, d_idx, nnz, dim, f1_sorted, &plan2, h_skew2, side))) return rc;
    if (direct) {
        // (eq(g, .) was built during the index check; the pass leaves eq(g,z) * v behind, per non-zero, in d_gv -- the list form's value buffer, unused on this route: phase two's input)
        d_a = d_gv;
        if ((rc = bucketed_dense<1>(mem, d_idx, d_vals, nnz, dim, eq_g, eq_u, d_f3, f1_sorted, d_hg, &direct, s, d_a))) return rc; // mod.rs:30-38
    }
    auto list_form = [&]() -> int {
        int r = sort_sparse(mem, d_idx, d_vals, nnz, 3 * dim, d_idx_s, d_vals_s, s);
        if (r) return r;
        lap("sort f1");
        have_list = true;
        return phase_one_device(mem, d_idx_s, d_vals_s, nnz, dim, d_f3, reinterpret_cast<const sch::Fr *>(g), d_hg, d_gi, d_gv, d_n1, &n1, s); // mod.rs:106
    };
    if (!direct && (rc = list_form())) return rc;
    std::vector<sch::Fr> u(dim), v(dim);
    ProverGuard pg;
    pg.take_cached(dim, mem.leased);
    // (phase one's sumcheck runs on the prover's stream, which is not ordered behind `s`: the host waits.  Ordering the prover's stream behind an
    // event on `s` instead was measured in round 6: the cross-queue dependency costs more than the wake-up it saves, 1.02 against 0.98 ms)
    G_TRY(hipStreamSynchronize(s));
    lap("phase one init");
    if ((rc = run_phase(rng->rng, &pg.p, d_hg, d_f2, dim, out_proof, u.data()))) return rc; // mod.rs:107-119
    lap("phase one sumcheck");
    // f2.evaluate(&u) (mod.rs:122) is not recomputed: phase one's prover has bound f2 at u_0..u_{dim-2} (prover.rs:84-89), its final table {lo, hi}
    // gives f2(u) = lo + u_last (hi - lo); and f3 is not multiplied by it (mod.rs:71-75): the scalar becomes phase two's coefficient, on the
    // device (sc_internal_scale_by_bound_table) -- the messages are the same canonical bits.  start_phase2_sumcheck / sc_dense_scale remain for callers.
    const void *f2_bound = sc_internal_bound_table(pg.p, 1);
    if (!f2_bound) return sc_internal_fail(SC_ERR_HIP, "phase one left no bound f2 table");
    // Phase two's initialisation goes onto the PROVER's stream, behind phase one's last kernel and in front of phase two's first: the eq tables,
    // [the terms,] the cells' sums, the coefficient -- no host synchronisation in between.
    const hipStream_t ps = sc_internal_prover_stream(pg.p);
    if (direct) {
        if (side != s) G_TRY(hipStreamSynchronize(side)); // (the plan: long finished, it ran beside phase one)
        direct = plan2.ok && *h_skew2 == 0;
    }
    if (direct) {
        if ((rc = build_eq_split(mem, u.data(), dim, &eq_u, ps))) return rc;
        if ((rc = bucket_fill<4>(plan2, d_idx, d_a, nnz, dim, eq_g, eq_u, d_f3, d_f1gu, ps))) return rc; // mod.rs:121: eq(u,x) * (eq(g,z) * v)
    } else {
        if (!have_list) { // (crowded y buckets although the x buckets were fine: build the list now)
            int r = sort_sparse(mem, d_idx, d_vals, nnz, 3 * dim, d_idx_s, d_vals_s, s);
            if (r) return r;
            if ((r = sparse_fix(mem, d_idx_s, d_vals_s, nnz, reinterpret_cast<const sch::Fr *>(g), dim, d_gi, d_gv, d_n1, s))) return r;
            unsigned int h_n1 = 0;
            G_TRY(hipMemcpyAsync(&h_n1, d_n1, sizeof(h_n1), hipMemcpyDeviceToHost, s));
            G_TRY(hipStreamSynchronize(s));
            n1 = h_n1;
            have_list = true;
        }
        if ((rc = phase_two_device(mem, d_gi, d_gv, n1, dim, u.data(), d_f1gu, s))) return rc; // mod.rs:121
        G_TRY(hipStreamSynchronize(s)); // (the prover's stream is not ordered behind `s`)
    }
    if (trace) {
        G_TRY(hipStreamSynchronize(ps));
        lap("phase two init");
    }
    const sch::Fr u_last = u[dim - 1];
    const std::function<int(sc_prover *)> coeff = [&](sc_prover *hp) -> int {
        scd::plan_hit(scd::kPlanGkrCoeffFromBound);
        return sc_internal_scale_by_bound_table(hp, f2_bound, u_last);
    };
    if ((rc = run_phase(rng->rng, &pg.p, d_f1gu, d_f3, dim, out_proof + (size_t)dim * 12, v.data(), &coeff))) return rc; // mod.rs:122-133
    lap("phase two sumcheck");
    if (out_uv_or_null) {
        std::memcpy(out_uv_or_null, u.data(), (size_t)dim * 32);
        std::memcpy(out_uv_or_null + (size_t)dim * 4, v.data(), (size_t)dim * 32);
    }
    return SC_OK;
}
