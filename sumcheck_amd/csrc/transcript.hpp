// transcript.hpp -- Blake2b512Rng: the Fiat-Shamir transcript of reference src/rng.rs:22-81, host side.
//
// BLAKE2b-512 (RFC 7693, unkeyed, 64-byte digest) is the `blake2` crate's Blake2b512.  The running
// digest absorbs serialised messages (`feed`) and is squeezed by finalising a CLONE of the state and
// re-absorbing each produced 64-byte block (`fill_bytes`).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "host_fr.hpp"

namespace sch {

class Blake2b512 {
  public:
    Blake2b512() { reset(); }
    void reset() {
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        for (int i = 0; i < 8; ++i) h_[i] = iv[i];
        h_[0] ^= 0x01010040ULL; // param block: digest_length=64, key_length=0, fanout=1, depth=1
        counter_lo_ = counter_hi_ = 0;
        fill_ = 0;
    }
    void update(const uint8_t *data, size_t len) {
        while (len) {
            if (fill_ == 128) { // a full block is only compressed once more input is known to follow
                bump(128);
                compress(block_, false);
                fill_ = 0;
            }
            size_t n = 128 - fill_;
            if (n > len) n = len;
            std::memcpy(block_ + fill_, data, n);
            fill_ += n;
            data += n;
            len -= n;
        }
    }
    // digest of everything absorbed so far; the object itself is left untouched (finalize on a clone)
    void digest(uint8_t out[64]) const {
        Blake2b512 c = *this;
        c.bump(c.fill_);
        std::memset(c.block_ + c.fill_, 0, 128 - c.fill_);
        c.compress(c.block_, true);
        std::memcpy(out, c.h_, 64);
    }

  private:
    static inline uint64_t ror(uint64_t x, unsigned n) { return (x >> n) | (x << (64 - n)); }
    void bump(uint64_t n) {
        counter_lo_ += n;
        if (counter_lo_ < n) ++counter_hi_;
    }
    void compress(const uint8_t *blk, bool last) {
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        static const uint8_t sigma[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint64_t m[16], v[16];
        std::memcpy(m, blk, 128); // little-endian host
        for (int i = 0; i < 8; ++i) {
            v[i] = h_[i];
            v[8 + i] = iv[i];
        }
        v[12] ^= counter_lo_;
        v[13] ^= counter_hi_;
        if (last) v[14] = ~v[14];
        auto mix = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] += v[b] + x; v[d] = ror(v[d] ^ v[a], 32);
            v[c] += v[d];     v[b] = ror(v[b] ^ v[c], 24);
            v[a] += v[b] + y; v[d] = ror(v[d] ^ v[a], 16);
            v[c] += v[d];     v[b] = ror(v[b] ^ v[c], 63);
        };
        for (int round = 0; round < 12; ++round) {
            const uint8_t *s = sigma[round % 10];
            mix(0, 4, 8, 12, m[s[0]], m[s[1]]);
            mix(1, 5, 9, 13, m[s[2]], m[s[3]]);
            mix(2, 6, 10, 14, m[s[4]], m[s[5]]);
            mix(3, 7, 11, 15, m[s[6]], m[s[7]]);
            mix(0, 5, 10, 15, m[s[8]], m[s[9]]);
            mix(1, 6, 11, 12, m[s[10]], m[s[11]]);
            mix(2, 7, 8, 13, m[s[12]], m[s[13]]);
            mix(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[8 + i];
    }
    uint64_t h_[8];
    uint64_t counter_lo_, counter_hi_;
    uint8_t block_[128];
    size_t fill_;
};

// FeedableRNG + RngCore of reference src/rng.rs
class Blake2b512Rng {
  public:
    // feed(): absorb an already-serialised message (rng.rs:36-41)
    void feed_bytes(const uint8_t *buf, size_t len) { digest_.update(buf, len); }
    // fill_bytes / try_fill_bytes (rng.rs:61-80)
    void fill_bytes(uint8_t *dest, size_t len) {
        uint8_t out[64];
        digest_.digest(out);
        size_t used = 0;
        for (size_t i = 0; i < len; ++i) {
            dest[i] = out[used++];
            if (used == 64) {
                digest_.update(out, 64);
                digest_.digest(out);
                used = 0;
            }
        }
        digest_.update(out, 64); // rng.rs:78: the current block is absorbed even if only partly used
    }
    uint64_t next_u64() { // rng.rs:51-55
        uint8_t t[8];
        fill_bytes(t, 8);
        uint64_t x;
        std::memcpy(&x, t, 8);
        return x;
    }
    // CanonicalSerialize of PolynomialInfo {max_multiplicands, num_variables} (data_structures.rs:47-55): 2 x u64 LE
    void feed_poly_info(uint64_t max_multiplicands, uint64_t num_variables) {
        uint8_t b[16];
        std::memcpy(b, &max_multiplicands, 8);
        std::memcpy(b + 8, &num_variables, 8);
        feed_bytes(b, 16);
    }
    // CanonicalSerialize of ProverMsg {evaluations: Vec<F>} (prover.rs:13-17): u64 LE length, then each element as
    // 32 bytes LE of its canonical (non-Montgomery) integer
    void feed_prover_msg(const Fr *evals, uint32_t n) {
        uint64_t len = n;
        feed_bytes(reinterpret_cast<const uint8_t *>(&len), 8);
        for (uint32_t i = 0; i < n; ++i) {
            const Fr c = to_canonical(evals[i]);
            feed_bytes(reinterpret_cast<const uint8_t *>(c.l), 32);
        }
    }
    // sample_round (verifier.rs:128-131) = F::rand: ark-ff's Fp sampler draws 4 x next_u64 as LE limbs, clears
    // the 256-255 = 1 unused top bit, rejects >= p; the accepted limbs ARE the Montgomery representation.
    Fr sample_fr() {
        for (;;) {
            Fr a;
            for (int i = 0; i < 4; ++i) a.l[i] = next_u64();
            a.l[3] &= 0xffffffffffffffffULL >> 1;
            if (!geq_p(a)) return a;
        }
    }

  private:
    Blake2b512 digest_;
};

} // namespace sch
