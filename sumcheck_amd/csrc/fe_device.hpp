// fe_device.hpp -- carry-free ("unsaturated") BLS12-381 Fr arithmetic for the big-round kernels on gfx950.
//
// Why a second representation.  On gfx950 a v_mad_u64_u32 costs the same issue time as a v_addc_co_u32
// (profiles/r1_instr_bench.txt: both 1.66x a v_add_u32), and a VALU carry needs two wait states before it can
// be consumed.  In the saturated 8 x 32-bit Montgomery product (fr_device.hpp) every one of the 128 multiply-adds
// drags a carry instruction behind it, and every modular add/sub is three serial carry chains.  Here an element
// is 9 signed limbs of 29 bits (radix 2^29, value = sum l_i 2^(29 i)): a column of the schoolbook product is at
// most 9 products below 2^59 plus 8 reduction products below 2^58, which fits a signed 64-bit accumulator, so
// the whole Montgomery product is 153 v_mad_i64_i32 with NO carry instruction, and add/sub are 9 independent
// v_add/v_sub (lazy: limbs may leave [0, 2^29), they are renormalised only where a bound requires it).
//
// SUBTRACTIVE Montgomery steps.  p = 1 (mod 2^29), so the textbook step adds m p with m = -acc mod 2^29 to clear the
// low limb.  Here the step SUBTRACTS m' p with m' = acc mod 2^29 (a single v_and): column k loses exactly its low 29
// bits, so "clear, then shift" is the arithmetic shift alone, and every reduction product m' * (-p_l) is a SIGNED
// multiply-add like the operand products -- one kind of instruction, one accumulator chain per column.  Measured in the
// ISA of one product: 153 mads + 17 v_lshl_add_u64 + 16 v_ashrrev_i64 + 17 v_and, against 153 + 27 + 16 + 17 + 9 v_sub
// + 8 v_mov for the additive form (the compiler kept the signed and the unsigned products in separate chains and paid a
// 64-bit add per column to join them, plus the 64-bit add of m itself).  The result is (T - M p) / 2^261 with
// 0 <= M < 2^261: congruent to T / 2^261 as before, in (T / 2^261 - p, T / 2^261] instead of [T / 2^261, T / 2^261 + p).
//
// Montgomery radix.  9 x 29 = 261, so fe_mul(a, b) = a*b / 2^261 (mod p), while tables hold the reference's
// R = 2^256 form.  Multiplying two R-form values therefore yields the R-form product times 2^-5.  The kernels
// compensate exactly: the bound challenge is pre-multiplied by 2^5 on the host (so r*(hi-lo) lands in R-form
// and can be added to `lo`), and a product of M multiplicands carries 2^(-5(M-1)), which is folded into that
// product's coefficient before k_finalize multiplies by it.  Field arithmetic is exact, so the round polynomial
// is bit-identical to the reference's.
//
// Bounds (|limb| of the two operands of fe_mul): one operand <= 2^30 (a normalised value plus one lazy add or
// sub), the other <= 2^29 (normalised, or a difference of two normalised values).  Then every column is below
// 9*2^59 + 8*2^58 + carry < 2^63 in magnitude.  Values (not limbs) stay below 2^259 in magnitude, results below 2^257 + p.
#pragma once
#include <utility>
#include "fr_device.hpp"
#include "kernels.h"

namespace scd {

struct Fe {
    int32_t l[9];
};
struct FeU { // wave-uniform (SGPR) element, normalised limbs
    int32_t l[9];
};

constexpr int32_t kFeMask = 0x1fffffff;

// p in radix 2^29
__device__ __forceinline__ constexpr int32_t fe_p_limb(int i) {
    return i == 0 ? 0x00000001 : i == 1 ? 0x1ffffff8 : i == 2 ? 0x1f96ffbf : i == 3 ? 0x1b4805ff : i == 4 ? 0x1d80553b
         : i == 5 ? 0x0c0404d0 : i == 6 ? 0x1520cce7 : i == 7 ? 0x0a6533af : 0x0073eda7;
}

// 8 x u32 (value < 2^256) -> 9 x 29-bit limbs, same value
__device__ __forceinline__ Fe fe_from_fr(const Fr &a) {
    // limb i holds bits 29i .. 29i+28 = (word i-1 >> (32 - 3i)) | (word i << 3i), i = 1..7; limb 8 = bits 232..255
    Fe r;
    r.l[0] = (int32_t)(a.v[0] & (uint32_t)kFeMask);
#pragma unroll
    for (int i = 1; i < 8; ++i) r.l[i] = (int32_t)(((a.v[i - 1] >> (32 - 3 * i)) | (a.v[i] << (3 * i))) & (uint32_t)kFeMask);
    r.l[8] = (int32_t)(a.v[7] >> 8);
    return r;
}
__device__ __forceinline__ FeU feu_from_host(const FrHost &h) {
    Fr a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a.v[2 * i] = (uint32_t)h.l[i];
        a.v[2 * i + 1] = (uint32_t)(h.l[i] >> 32);
    }
    const Fe e = fe_from_fr(a);
    FeU u;
#pragma unroll
    for (int i = 0; i < 9; ++i) u.l[i] = e.l[i];
    return u;
}

// 32 * a as nine 29-bit limbs, a < 2^256 given as 8 x u32 (uniform): the challenge as the carry-free bind takes it (fe_mul divides by
// 2^261, the tables hold the R = 2^256 form, so r * 2^5 makes r * (hi - lo) land in R-form).  The value 32 a < 2^261 is NOT reduced
// mod p: as a multiplier it only has to have limbs below 2^29.
__device__ __forceinline__ FeU feu_shl5(const uint32_t (&v)[8]) {
    FeU u;
    u.l[0] = (int32_t)((v[0] << 5) & (uint32_t)kFeMask);
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int s = 29 * i - 5, w = s >> 5, off = s & 31; // bits [s, s + 29) of a
        uint32_t x = v[w] >> off;
        if (off > 3 && w + 1 < 8) x |= v[w + 1] << (32 - off);
        u.l[i] = (int32_t)(x & (uint32_t)kFeMask);
    }
    return u;
}

__device__ __forceinline__ Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
}
__device__ __forceinline__ Fe fe_add(const Fe &a, const Fe &b) { // lazy: no carry propagation
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
__device__ __forceinline__ Fe fe_sub(const Fe &a, const Fe &b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i];
    return r;
}
// carry propagation: limbs 0..7 into [0, 2^29), limb 8 keeps the (signed) rest.  Value unchanged.
__device__ __forceinline__ Fe fe_normalize(const Fe &a) {
    Fe r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int32_t t = a.l[i] + c;
        r.l[i] = t & kFeMask;
        c = t >> 29; // arithmetic
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// One parallel carry-save pass: every limb sheds its bits above 2^29 into the next limb at once (no ripple, three
// independent ops per limb).  For |limbs| < 2^31 the result has limbs 0..7 in [-4, 2^29 + 4): as good as normalised
// for the product bounds, at a latency of three instructions instead of a 24-deep dependent chain.  Value unchanged.
__device__ __forceinline__ Fe fe_carry_pass(const Fe &a) {
    Fe r;
    r.l[0] = a.l[0] & kFeMask;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.l[i] = (a.l[i] & kFeMask) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    return r;
}

// ---- internal table format F29 -----------------------------------------------------------------------------------
// An element is nine 29-bit limbs.  Limb 8 lives in a linear int32 array top[entry].  Limbs 0..7 live in the "main" array in
// a chunk-planar layout chosen so that every store instruction of the bind kernel is contiguous across the wavefront:
// entries are grouped in blocks of 64 pairs (128 entries, 4 KiB); inside a block, plane k = 2*(entry & 1) + half holds the
// 16-byte chunk (limbs 4*half .. 4*half+3) of that entry for the 64 pairs side by side:
//     byte offset(entry e, half) = (e >> 7) * 4096 + (2 * (e & 1) + half) * 1024 + col(q) * 16,  q = e >> 1 (the pair),
//     col(q) = ((q & 63) >> 1) | ((q & 1) << 5)   -- even pairs of the block in columns 0..31, odd pairs in 32..63.
// The writer lane of pair q stores plane k at column col(q): the 64 lanes of a wavefront cover one contiguous 1 KiB row per
// instruction (the reference layout's lane-private 64-byte stores cost 12 % of the mixed read/write HBM rate,
// profiles/r1_mem_bench.txt).  The next round's lane b reads pairs 2b and 2b+1, i.e. for each of them 32 consecutive columns
// per half-wavefront: every load instruction is two contiguous 512-byte runs.
// Tables in this format always hold a multiple of 128 entries (big rounds only).
// Both directions are STREAMED (nontemporal): a big round reads every element of its source tables once and writes every element of its
// destination tables once, fully coalesced in this layout.  Measured together (profiles/r3y_nontemporal_ab.txt): -2.4 % per proof, rounds 3-6
// -3...-10 %; the stores alone -1.1 %, the loads alone nothing.  (The caller's canonical tables are NOT read this way: their 16-byte-per-lane
// loads at a 64 / 128-byte stride live off the L1 hits that the hint gives up -- rounds 1 and 2 were 19 % and 69 % slower.)
#ifndef SC_NO_NT // (A/B build without the hints)
typedef uint32_t sc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 f29_ld(const uint4 *p) {
    const sc_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const sc_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void f29_st(uint4 *p, const uint32_t a, const uint32_t b, const uint32_t c, const uint32_t d) {
    sc_u32x4 w;
    w.x = a; w.y = b; w.z = c; w.w = d;
    __builtin_nontemporal_store(w, reinterpret_cast<sc_u32x4 *>(p));
}
#else
__device__ __forceinline__ uint4 f29_ld(const uint4 *p) { return *p; }
__device__ __forceinline__ void f29_st(uint4 *p, const uint32_t a, const uint32_t b, const uint32_t c, const uint32_t d) { *p = make_uint4(a, b, c, d); }
#endif
__device__ __forceinline__ uint64_t f29_chunk(uint64_t entry, int half) { // index in uint4 units
    const uint64_t q = entry >> 1; // pair
    const uint64_t col = ((q & 63) >> 1) | ((q & 1) << 5); // even pairs in columns 0..31, odd pairs in 32..63
    return (entry >> 7) * 256 + (uint64_t)(2 * (int)(entry & 1) + half) * 64 + col;
}
__device__ __forceinline__ Fe fe_load_f29(const uint4 *main, uint64_t entry, int32_t top) {
    const uint4 a = f29_ld(main + f29_chunk(entry, 0)), b = f29_ld(main + f29_chunk(entry, 1));
    Fe r;
    r.l[0] = (int32_t)a.x; r.l[1] = (int32_t)a.y; r.l[2] = (int32_t)a.z; r.l[3] = (int32_t)a.w;
    r.l[4] = (int32_t)b.x; r.l[5] = (int32_t)b.y; r.l[6] = (int32_t)b.z; r.l[7] = (int32_t)b.w;
    r.l[8] = top;
    return r;
}
__device__ __forceinline__ void fe_store_f29(uint4 *main, uint64_t entry, const Fe &v) {
    f29_st(main + f29_chunk(entry, 0), (uint32_t)v.l[0], (uint32_t)v.l[1], (uint32_t)v.l[2], (uint32_t)v.l[3]);
    f29_st(main + f29_chunk(entry, 1), (uint32_t)v.l[4], (uint32_t)v.l[5], (uint32_t)v.l[6], (uint32_t)v.l[7]);
}
// One multiply-add of a column.  kChain: the instruction is written out, so that every multiply-add of a product accumulates into ONE
// register pair in program order -- the compiler otherwise starts each column's chain from zero and joins it to the carry with a 64-bit
// add (17 v_lshl_add_u64 per product).  A dependent v_mad_i64_i32 issues back to back (tools/instr_bench.hip: one dependent chain on
// one wavefront per SIMD runs at 87 % of the saturated rate, at two wavefronts at 100 %), so the chain itself costs nothing, and it needs
// fewer registers (round 1: 158 instead of 168 + 28 bytes of scratch).  Measured per round (profiles/r3g_ab.txt): the kernels that read
// canonical tables gain (round 1 -5.6 %, round 2 -2.9 %); the rounds that stream the internal format lose 2-6 % (their few instructions
// per byte leave the memory pipe in charge, and the independent column chains give the scheduler more to overlap with it) -- so the chain
// is a template parameter: on for round 1 and for the first binding round, off elsewhere.
template <bool kChain>
__device__ __forceinline__ void fe_mad(int64_t &acc, const int32_t a, const int32_t b) {
    if constexpr (kChain) asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    else acc += (int64_t)a * (int64_t)b;
}
template <bool kChain>
__device__ __forceinline__ void fe_mad_k(int64_t &acc, const int32_t a, const int32_t k) { // k: a compile-time constant (an SGPR)
    if constexpr (kChain) asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
    else acc += (int64_t)a * (int64_t)k;
}
#include "fe_mad_chain.inc" // mad_chain_vv / mad_chain_vs: a column's multiply-adds as ONE asm statement (tools/gen_mad_chain.py)
#ifdef SC_CHAIN_PER_MAD // A/B build: one asm statement per multiply-add (each followed by the hazard recogniser's s_nop)
constexpr bool kChainColumns = false;
#else
constexpr bool kChainColumns = true;
#endif
#ifdef SC_MAD_CHAIN // A/B build: the written-out chain everywhere
constexpr bool kChainDefault = true;
#else
constexpr bool kChainDefault = false;
#endif

// Column k of a 9 x 9 limb product: its operand products a_i b_{k-i} and its reduction products m_j (-p_{k-j}), as compile-time
// counts -- the written-out chains (kChain) issue a column as ONE asm statement selected by these (MadMix, fe_mad_chain.inc).
__host__ __device__ constexpr int fe_col_n(int k) { return (k < 8 ? k : 8) - (k > 8 ? k - 8 : 0) + 1; }
__host__ __device__ constexpr int fe_col_nr(int k) { // j < k, 1 <= k - j <= 8, j <= 8
    int c = 0;
    for (int j = 0; j < 9; ++j) c += (j < k && k - j >= 1 && k - j < 9) ? 1 : 0;
    return c;
}
template <int K, typename B>
__device__ __forceinline__ void fe_col_chain(int64_t &acc, const Fe &a, const B &b, const int32_t (&m)[9]) {
    constexpr int n = fe_col_n(K), nr = fe_col_nr(K);
    int32_t xa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int q = 0, qr = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int j = K - i;
        if (j >= 0 && j < 9) { xa[q] = a.l[i]; xb[q] = b.l[j]; ++q; }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int l = K - j;
        if (j < K && l >= 1 && l < 9) { xm[qr] = m[j]; xk[qr] = -fe_p_limb(l); ++qr; }
    }
    if constexpr (nr == 0) mad_chain_vv(acc, xa, xb, n);
    else MadMix<n, nr>::run(acc, xa, xb, xm, xk);
}
// the same for (a b + c d): the a b products as one statement, the c d products with the reduction products as another
template <int K>
__device__ __forceinline__ void fe_col2_chain(int64_t &acc, const Fe &a, const Fe &b, const Fe &c, const Fe &d, const int32_t (&m)[9]) {
    constexpr int n = fe_col_n(K), nr = fe_col_nr(K);
    int32_t xa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int32_t xm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int q = 0, qr = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int j = K - i;
        if (j >= 0 && j < 9) { xa[q] = a.l[i]; xb[q] = b.l[j]; xc[q] = c.l[i]; xd[q] = d.l[j]; ++q; }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int l = K - j;
        if (j < K && l >= 1 && l < 9) { xm[qr] = m[j]; xk[qr] = -fe_p_limb(l); ++qr; }
    }
    mad_chain_vv(acc, xa, xb, n);
    if constexpr (nr == 0) mad_chain_vv(acc, xc, xd, n);
    else MadMix<n, nr>::run(acc, xc, xd, xm, xk);
}
// the Montgomery bookkeeping between columns (subtractive steps: the low 29 bits are the next multiplier, or a limb of the result)
template <int K>
__device__ __forceinline__ void fe_col_finish(int64_t &acc, int32_t (&m)[9], Fe &r) {
    if constexpr (K < 9) m[K] = (int32_t)((uint32_t)acc & (uint32_t)kFeMask);
    else r.l[K - 9] = (int32_t)((uint32_t)acc & (uint32_t)kFeMask);
    acc >>= 29;
}
template <typename B, int... K>
__device__ __forceinline__ Fe fe_mul_chain_cols(const Fe &a, const B &b, std::integer_sequence<int, K...>) {
    int64_t acc = 0;
    int32_t m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Fe r;
    ((fe_col_chain<K, B>(acc, a, b, m), fe_col_finish<K>(acc, m, r)), ...);
    r.l[8] = (int32_t)acc;
    return r;
}
template <int... K>
__device__ __forceinline__ Fe fe_mul2_chain_cols(const Fe &a, const Fe &b, const Fe &c, const Fe &d, std::integer_sequence<int, K...>) {
    int64_t acc = 0;
    int32_t m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Fe r;
    ((fe_col2_chain<K>(acc, a, b, c, d, m), fe_col_finish<K>(acc, m, r)), ...);
    r.l[8] = (int32_t)acc;
    return r;
}

// a * b / 2^261 (mod p), result value in (a b / 2^261 - p, a b / 2^261], i.e. |.| < 2^257 + p; result limbs 0..7 in [0, 2^29), limb 8 signed and small.
template <typename B, bool kChain = kChainDefault>
__device__ __forceinline__ Fe fe_mul_t(const Fe &a, const B &b) {
    if constexpr (kChain && kChainColumns) return fe_mul_chain_cols<B>(a, b, std::make_integer_sequence<int, 17>{});
    int64_t acc = 0;
    int32_t m[9];
    Fe r;
#pragma unroll
    for (int k = 0; k <= 16; ++k) {
        {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int j = k - i;
            if (j >= 0 && j < 9) fe_mad<kChain>(acc, a.l[i], b.l[j]);
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int l = k - j;
            if (j < k && l >= 1 && l < 9) fe_mad_k<kChain>(acc, m[j], -fe_p_limb(l));
        }
        }
        if (k < 9) {
            m[k] = (int32_t)((uint32_t)acc & (uint32_t)kFeMask); // subtracting m p_0 = m clears the low 29 bits: that IS the shift below
        } else {
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)kFeMask);
        }
        acc >>= 29; // arithmetic shift = floor: drops the low limb (k < 9: subtracted, see above; k >= 9: just extracted)
    }
    r.l[8] = (int32_t)acc;
    return r;
}
template <bool kChain = kChainDefault>
__device__ __forceinline__ Fe fe_mul(const Fe &a, const Fe &b) { return fe_mul_t<Fe, kChain>(a, b); }
// (a * b + c * d) / 2^261 (mod p) with ONE Montgomery reduction: 162 + 72 multiply-adds instead of 2 x 153.  For sums that are only
// accumulated (the final products of two pairs of the same evaluation node).  Bounds: all four operands |limb| <= 2^29 + 4, so a
// column is within 18 * 2^58.01 + 8 * 2^58 + carry < 2^63 in magnitude.
template <bool kChain = kChainDefault>
__device__ __forceinline__ Fe fe_mul2_sum(const Fe &a, const Fe &b, const Fe &c, const Fe &d) {
    if constexpr (kChain && kChainColumns) return fe_mul2_chain_cols(a, b, c, d, std::make_integer_sequence<int, 17>{});
    int64_t acc = 0;
    int32_t m[9];
    Fe r;
#pragma unroll
    for (int k = 0; k <= 16; ++k) {
        {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int j = k - i;
            if (j >= 0 && j < 9) {
                fe_mad<kChain>(acc, a.l[i], b.l[j]);
                fe_mad<kChain>(acc, c.l[i], d.l[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int l = k - j;
            if (j < k && l >= 1 && l < 9) fe_mad_k<kChain>(acc, m[j], -fe_p_limb(l));
        }
        }
        if (k < 9) {
            m[k] = (int32_t)((uint32_t)acc & (uint32_t)kFeMask);
        } else {
            r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)kFeMask);
        }
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    return r;
}
template <bool kChain = kChainDefault>
__device__ __forceinline__ Fe fe_mul_u(const Fe &a, const FeU &u) { return fe_mul_t<FeU, kChain>(a, u); }

// d * r (mod p) for the round's fixed challenge r, d the lazy difference of two table entries (|limbs| < 2^29 + 16).
// C.R[i] = (r * 2^(29 i + 58)) mod p, so S = sum_i d_i * R_i is congruent to d * r * 2^58 and already NINE columns wide: no high
// half to fold.  Two (subtractive) Montgomery steps (m0, m1; p_0 = 1) divide by 2^58:
//   T = (S - m0 p - m1 p 2^29) / 2^58,   |T| < 2^230 + p (1 + 2^-29)   (T in (-p - 2^230, 2^230)),
// 81 + 16 = 97 multiply-adds instead of the 153 of a general product.  Result limbs 0..7 in [0, 2^29), limb 8 signed, |.| < 2^24.
// The 81 constants do not fit the SGPR file next to everything else, so the block parks them in LDS, column-major
// (RT[12 k + i] = R[i][k], kBindLds ints), and every column's nine constants come back with three broadcast ds_read_b128 that
// serve BOTH products of a table's pair (entries 2b and 2b+1 are bound by the same r).
constexpr int kBindLds = 9 * 12;
__device__ __forceinline__ void bind_consts_to_lds(const BindConst &C, int32_t (&RT)[kBindLds]) {
    for (int i = threadIdx.x; i < kBindLds; i += blockDim.x) {
        const int k = i / 12, row = i % 12;
        RT[i] = row < 9 ? C.R[row][k] : 0;
    }
    __syncthreads();
}
template <bool kChain = kChainDefault>
__device__ __forceinline__ Fe fe_mul_bind(const Fe &d, const int32_t (&RT)[kBindLds]) {
    uint32_t off = 0;
    asm volatile("" : "+v"(off)); // keep the constant loads inside the pair loop (hoisted, they would pin 81 VGPRs)
    const int32_t *q = RT + off;
    int64_t acc = 0;
    int32_t m0 = 0, m1 = 0;
    Fe r;
    // one column ahead: column k+1's constants are in flight while column k multiplies; the compiler barriers keep the 27
    // loads from being issued up front (72 more live VGPRs)
    int4 n0 = *reinterpret_cast<const int4 *>(q), n1 = *reinterpret_cast<const int4 *>(q + 4);
    int32_t n2 = q[8];
#pragma unroll
    for (int k = 0; k <= 9; ++k) {
        if (k < 9) {
            const int32_t c[9] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2};
            asm volatile("" : "+v"(acc)::"memory"); // orders column k-1's multiply-adds before column k+1's loads
            if (k < 8) {
                n0 = *reinterpret_cast<const int4 *>(q + 12 * (k + 1));
                n1 = *reinterpret_cast<const int4 *>(q + 12 * (k + 1) + 4);
                n2 = q[12 * (k + 1) + 8];
            }
            if constexpr (kChain && kChainColumns) { // the column's nine products and its one or two reduction products as one statement
                const int32_t xd[9] = {d.l[0], d.l[1], d.l[2], d.l[3], d.l[4], d.l[5], d.l[6], d.l[7], d.l[8]};
                int32_t xm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                int nr = 0;
                if (k >= 1) { xm[nr] = m0; xk[nr] = -fe_p_limb(k); ++nr; }
                if (k >= 2) { xm[nr] = m1; xk[nr] = -fe_p_limb(k - 1); ++nr; }
                if (nr == 0) mad_chain_vv(acc, xd, c, 9);
                else if (nr == 1) MadMix<9, 1>::run(acc, xd, c, xm, xk);
                else MadMix<9, 2>::run(acc, xd, c, xm, xk);
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) fe_mad<kChain>(acc, d.l[i], c[i]);
            }
        }
        if constexpr (!(kChain && kChainColumns)) {
            if (k >= 1 && k < 9) fe_mad_k<kChain>(acc, m0, -fe_p_limb(k));
        }
        if (!(kChain && kChainColumns) || k == 9) {
            if (k >= 2) fe_mad_k<kChain>(acc, m1, -fe_p_limb(k - 1));
        }
        if (k == 0) {
            m0 = (int32_t)((uint32_t)acc & (uint32_t)kFeMask); // subtractive steps, as in fe_mul_t
        } else if (k == 1) {
            m1 = (int32_t)((uint32_t)acc & (uint32_t)kFeMask);
        } else {
            r.l[k - 2] = (int32_t)((uint32_t)acc & (uint32_t)kFeMask);
        }
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    return r;
}

// Compiler fence on three elements: they are computed before, and stay in registers across, this point.  (Without it the
// half-products of factors 0,1 are sunk below the loads and binds of factors 2,3 and all four lines are live at once.)
__device__ __forceinline__ void fe_pin3(Fe &a, Fe &b, Fe &c) {
    asm volatile(""
                 : "+v"(a.l[0]), "+v"(a.l[1]), "+v"(a.l[2]), "+v"(a.l[3]), "+v"(a.l[4]), "+v"(a.l[5]), "+v"(a.l[6]), "+v"(a.l[7]), "+v"(a.l[8]),
                   "+v"(b.l[0]), "+v"(b.l[1]), "+v"(b.l[2]), "+v"(b.l[3]), "+v"(b.l[4]), "+v"(b.l[5]), "+v"(b.l[6]), "+v"(b.l[7]), "+v"(b.l[8]),
                   "+v"(c.l[0]), "+v"(c.l[1]), "+v"(c.l[2]), "+v"(c.l[3]), "+v"(c.l[4]), "+v"(c.l[5]), "+v"(c.l[6]), "+v"(c.l[7]), "+v"(c.l[8])
                 :
                 : "memory");
}

// Exact canonical conversion: any value with |v| < 2^260 -> the representative in [0, p) as 8 x u32.
__device__ __forceinline__ Fr fe_to_fr(const Fe &a) {
    // 1. normalise, estimate the quotient by p from the top limb (= floor(v / 2^232))
    Fe n = fe_normalize(a);
    constexpr int32_t PH = 0x0073eda7; // floor(p / 2^232)
    // q = floor(top / (PH + 1)) for top >= 0, and -(floor((-top - 1) / PH) + 1) for top < 0, by float reciprocal
    // (|top| < 2^28; the estimate only has to leave a remainder in [0, 2p), the exact step below finishes).
    const int32_t top = n.l[8];
    int32_t q;
    if (top >= 0) q = (int32_t)((uint32_t)top / (uint32_t)(PH + 1));
    else q = -(int32_t)(((uint32_t)(-top) + (uint32_t)PH - 1u) / (uint32_t)PH);
    // 2. v -= q * p  (64-bit per limb, then carry-normalise)
    int64_t c = 0;
    uint32_t w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int64_t t = (int64_t)n.l[i] - (int64_t)q * (int64_t)fe_p_limb(i) + c;
        w[i] = (uint32_t)t & (uint32_t)kFeMask;
        c = t >> 29;
        if (i == 8) w[8] = (uint32_t)t; // top limb keeps everything (now in [0, 2^25))
    }
    // 3. pack 9 x 29 -> 8 x 32 (value now in [0, 2p + small) < 2^256): word k = (limb k >> 3k) | (limb k+1 << (29 - 3k))
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] = (w[k] >> (3 * k)) | (w[k + 1] << (29 - 3 * k));
    // 4. exact: at most two conditional subtractions
    return fr_reduce_once(fr_reduce_once(r));
}

} // namespace scd
