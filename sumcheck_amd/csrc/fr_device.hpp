// fr_device.hpp -- BLS12-381 scalar-field arithmetic for gfx950 (CDNA4) device code.
//
// Replaces ark_ff::Fp<MontBackend<FrConfig,4>,4> mul/add/sub as used by the reference at
// src/ml_sumcheck/protocol/prover.rs:116,120,122,123,127,145.  Elements are 8 x u32 little-endian
// limbs in VGPRs, Montgomery form with R = 2^256 (identical bit pattern to ark-ff's 4 x u64), and
// every value that leaves a kernel is canonical (< p) so results are bit-exact with the reference.
//
// CDNA4 has no 64x64 multiplier; the widest integer multiply is v_mad_u64_u32 (32x32+64 -> 64), so
// the Montgomery product is a 32-bit-limb CIOS.  Two properties of this modulus are exploited:
//   * -p^-1 mod 2^32 = 0xffffffff, so the Montgomery quotient digit is m = -t0 (no multiply);
//   * p mod 2^32 = 1, so m*p0 + t0 = 2^32*[t0 != 0]: the first reduction column is a compare.
// The top limb of p is < 2^31 ("spare bit"): a+b never overflows 256 bits and the CIOS running value
// stays < 2p, so the interleaved form needs no 10th limb.
//
// fr_mul is the generated Comba/FIPS product (fr_mul_gen.inc, tools/gen_mac.py); fr_mul_cios is kept as
// the readable restatement it is tested against.  The big-round kernels do not use this file's product:
// they compute in the carry-free 9 x 29-bit representation of fe_device.hpp and only convert at the edges.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scd {

struct Fr {
    uint32_t v[8];
};

// p, 32-bit limbs, little-endian
#define SC_P0 0x00000001u
#define SC_P1 0xffffffffu
#define SC_P2 0xfffe5bfeu
#define SC_P3 0x53bda402u
#define SC_P4 0x09a1d805u
#define SC_P5 0x3339d808u
#define SC_P6 0x299d7d48u
#define SC_P7 0x73eda753u

__device__ __forceinline__ constexpr uint32_t fr_p_limb(int i) {
    return i == 0 ? SC_P0 : i == 1 ? SC_P1 : i == 2 ? SC_P2 : i == 3 ? SC_P3 : i == 4 ? SC_P4 : i == 5 ? SC_P5 : i == 6 ? SC_P6 : SC_P7;
}
// R^2 mod p (to_mont multiplier), 32-bit limbs
__device__ __forceinline__ constexpr uint32_t fr_r2_limb(int i) {
    return i == 0 ? 0xf3f29c6du : i == 1 ? 0xc999e990u : i == 2 ? 0x87925c23u : i == 3 ? 0x2b6cedcbu
         : i == 4 ? 0x7254398fu : i == 5 ? 0x05d31496u : i == 6 ? 0x9f59ff11u : 0x0748d9d9u;
}
// R mod p = Montgomery form of 1
__device__ __forceinline__ constexpr uint32_t fr_one_limb(int i) {
    return i == 0 ? 0xfffffffeu : i == 1 ? 0x00000001u : i == 2 ? 0x00034802u : i == 3 ? 0x5884b7fau
         : i == 4 ? 0xecbc4ff5u : i == 5 ? 0x998c4fefu : i == 6 ? 0xacc5056fu : 0x1824b159u;
}

__device__ __forceinline__ Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
}
__device__ __forceinline__ Fr fr_one() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = fr_one_limb(i);
    return r;
}

__device__ __forceinline__ Fr fr_load(const uint4 *p) {
    uint4 a = p[0], b = p[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void fr_store(uint4 *p, const Fr &a) {
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// r = a - p if a >= p else a   (a < 2p)
__device__ __forceinline__ Fr fr_reduce_once(const Fr &a) {
    Fr s;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s.v[i] = __builtin_subc(a.v[i], fr_p_limb(i), br, &br);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : s.v[i];
    return r;
}

__device__ __forceinline__ Fr fr_add(const Fr &a, const Fr &b) {
    Fr t;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    return fr_reduce_once(t); // spare top bit: no carry out of limb 7
}

__device__ __forceinline__ Fr fr_sub(const Fr &a, const Fr &b) {
    Fr t;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
    const uint32_t mask = 0u - br; // all ones if a < b
    Fr r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = __builtin_addc(t.v[i], fr_p_limb(i) & mask, c, &c);
    return r;
}

// Montgomery product a*b*R^-1 mod p, canonical output.  Interleaved (CIOS) form, 32-bit limbs, plain C++:
// the compiler's rendering spends ~3 v_mov + one 64-bit add per v_mad_u64_u32 (kept as the cross-check
// implementation; the production product is fr_mul_comba below).
__device__ __forceinline__ Fr fr_mul_cios(const Fr &a, const Fr &b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // t += a * b[i]
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t x = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        const uint32_t t8 = (uint32_t)c;
        // m = -t0 ; (t + m*p) >> 32
        const uint32_t m = 0u - t[0];
        c = (t[0] != 0u) ? 1u : 0u; // column 0: t0 + m*1 = 2^32 * [t0 != 0]
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            uint64_t x = (uint64_t)m * fr_p_limb(j) + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        t[7] = t8 + (uint32_t)c; // running value < 2p < 2^256: cannot overflow
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return fr_reduce_once(r);
}

// A wave-uniform element (e.g. the round challenge, a kernel argument): limbs live in SGPRs.
struct FrU {
    uint32_t v[8];
};

#include "fr_mac.inc"
#include "fr_mul_gen.inc"

#ifndef SC_MUL_IMPL
#define SC_MUL_IMPL 1
#endif
__device__ __forceinline__ Fr fr_mul(const Fr &a, const Fr &b) {
#if SC_MUL_IMPL == 0
    return fr_mul_cios(a, b);
#else
    return fr_mul_comba(a, b);
#endif
}
// a * u with u wave-uniform
__device__ __forceinline__ Fr fr_mul_u(const Fr &a, const FrU &u) {
#if SC_MUL_IMPL == 0
    Fr b;
#pragma unroll
    for (int i = 0; i < 8; ++i) b.v[i] = u.v[i];
    return fr_mul_cios(a, b);
#else
    return fr_mul_comba_u(a, u);
#endif
}

// small integer -> Montgomery form
__device__ __forceinline__ Fr fr_from_u32(uint32_t x) {
    Fr a = fr_zero(), r2;
    a.v[0] = x;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.v[i] = fr_r2_limb(i);
    return fr_mul(a, r2);
}

__device__ __forceinline__ Fr fr_neg(const Fr &a) { return fr_sub(fr_zero(), a); }

__device__ __forceinline__ bool fr_is_zero(const Fr &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}

} // namespace scd
