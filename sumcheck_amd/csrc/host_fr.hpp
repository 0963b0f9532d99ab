// host_fr.hpp -- host-side BLS12-381 Fr (4 x u64 Montgomery limbs) for the parts of the path that the
// reference also runs scalar on the host: transcript serialisation (canonical bytes), F::rand,
// the verifier's interpolation, and folding an integer all-reduce back into the field.
// All prove_round arithmetic runs on the GPU (kernels.hip); nothing here touches evaluation tables.
#pragma once
#include <cstdint>
#include <cstring>

namespace sch {

typedef unsigned __int128 u128;

struct Fr {
    uint64_t l[4];
};

static constexpr Fr kP = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
static constexpr Fr kOne = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}}; // R mod p
static constexpr Fr kR2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
static constexpr uint64_t kInv = 0xfffffffeffffffffULL; // -p^-1 mod 2^64

inline bool geq_p(const Fr &a) {
    for (int i = 3; i >= 0; --i) {
        if (a.l[i] != kP.l[i]) return a.l[i] > kP.l[i];
    }
    return true;
}
inline Fr sub_p(const Fr &a) {
    Fr r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - kP.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return r;
}
inline Fr zero() { return Fr{{0, 0, 0, 0}}; }
inline bool eq(const Fr &a, const Fr &b) { return std::memcmp(&a, &b, sizeof(Fr)) == 0; }

inline Fr add(const Fr &a, const Fr &b) {
    Fr r;
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    return geq_p(r) ? sub_p(r) : r;
}
inline Fr sub(const Fr &a, const Fr &b) {
    Fr r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)r.l[i] + kP.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
inline Fr neg(const Fr &a) { return sub(zero(), a); }

// Montgomery product, separated multiply-then-reduce form (SOS)
inline Fr mul(const Fr &a, const Fr &b) {
    uint64_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            carry += (u128)a.l[i] * b.l[j] + w[i + j];
            w[i + j] = (uint64_t)carry;
            carry >>= 64;
        }
        w[i + 4] = (uint64_t)carry;
    }
    for (int i = 0; i < 4; ++i) {
        const uint64_t q = w[i] * kInv;
        u128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            carry += (u128)q * kP.l[j] + w[i + j];
            w[i + j] = (uint64_t)carry;
            carry >>= 64;
        }
        for (int j = i + 4; carry != 0 && j < 9; ++j) {
            carry += w[j];
            w[j] = (uint64_t)carry;
            carry >>= 64;
        }
    }
    Fr r = {{w[4], w[5], w[6], w[7]}};
    return (w[8] != 0 || geq_p(r)) ? sub_p(r) : r;
}

inline Fr from_u64(uint64_t x) { return mul(Fr{{x, 0, 0, 0}}, kR2); }
inline Fr to_canonical(const Fr &mont) { return mul(mont, Fr{{1, 0, 0, 0}}); } // into_bigint()
inline Fr pow(const Fr &a, const uint64_t e[4]) {
    Fr r = kOne;
    for (int i = 255; i >= 0; --i) {
        r = mul(r, r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a);
    }
    return r;
}
inline Fr inverse(const Fr &a) {
    const uint64_t e[4] = {kP.l[0] - 2, kP.l[1], kP.l[2], kP.l[3]};
    return pow(a, e);
}

} // namespace sch
