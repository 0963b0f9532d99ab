// fs_device_bench.hip -- what would one round of the Fiat-Shamir transcript cost ON THE DEVICE?
//
// The transcript of reference src/rng.rs:36-41,61-80 + verifier.rs:128-131 (feed the round message, squeeze F::rand) is a strictly
// serial chain of BLAKE2b compressions: per round 8 + 32 D bytes of message are absorbed, then each of the four next_u64 finalises a
// clone of the state (one compression) and absorbs the 64-byte block it produced -- about 7 compressions per round, each depending on
// the one before it.  This program measures that chain on one MI355X in the two forms a device-side transcript could take, against
// the host's (transcript.hpp), bit for bit:
//   variant 0   one lane: the plain 16-word state in registers
//   variant 1   four lanes: lane j holds column j of the 4 x 4 state, the diagonal step rotates rows with DPP quad_perm, message words
//               come from LDS (the form every SIMD BLAKE2b takes; there is no more parallelism in the function than these four G's)
// Each runs R rounds of [feed_prover_msg(D elements), sample_fr] where round i's message is derived from round i-1's challenge (as in
// the protocol: nothing can be precomputed), in ONE wavefront of an otherwise idle GPU, timed with the constant 100 MHz clock inside the
// kernel and with events around it.
//
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sumcheck_amd/csrc tools/fs_device_bench.hip -o tools/fs_device_bench.bin && tools/fs_device_bench.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "transcript.hpp"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));          \
            return 2;                                                                              \
        }                                                                                          \
    } while (0)

__constant__ uint64_t c_iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
__constant__ uint8_t c_sigma[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

// 64-bit rotates as two v_alignbit_b32 (32: a register swap); the compiler's own lowering of the shift/or form takes four to six instructions
template <unsigned N>
__device__ __forceinline__ uint64_t ror64c(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (N == 32) return ((uint64_t)lo << 32) | hi;
    if (N < 32) return ((uint64_t)__builtin_amdgcn_alignbit(lo, hi, N) << 32) | __builtin_amdgcn_alignbit(hi, lo, N);
    return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, N - 32) << 32) | __builtin_amdgcn_alignbit(lo, hi, N - 32);
}
#define ror64(x, n) ror64c<n>(x)

// ---- variant 0: one lane -----------------------------------------------------------------------------------------------
struct B2State {
    uint64_t h[8];
    uint64_t t; // bytes compressed so far (the transcript of a proof stays far below 2^64)
    uint32_t fill;
};

#define G1(a, b, c, d, x, y)                \
    a += b + (x); d = ror64(d ^ a, 32);     \
    c += d;       b = ror64(b ^ c, 24);     \
    a += b + (y); d = ror64(d ^ a, 16);     \
    c += d;       b = ror64(b ^ c, 63);

__device__ __forceinline__ void compress1(uint64_t h[8], const uint64_t *m_lds, uint64_t t, bool last) {
    uint64_t m[16]; // the buffer lives in LDS (it is filled word by word at a run-time index); a compression reads it once
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = m_lds[i];
    uint64_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint64_t v8 = c_iv[0], v9 = c_iv[1], v10 = c_iv[2], v11 = c_iv[3], v12 = c_iv[4] ^ t, v13 = c_iv[5], v14 = last ? ~c_iv[6] : c_iv[6], v15 = c_iv[7];
#define ROUND1(r)                                                                                          \
    G1(v0, v4, v8, v12, m[SG[r][0]], m[SG[r][1]]) G1(v1, v5, v9, v13, m[SG[r][2]], m[SG[r][3]])            \
    G1(v2, v6, v10, v14, m[SG[r][4]], m[SG[r][5]]) G1(v3, v7, v11, v15, m[SG[r][6]], m[SG[r][7]])          \
    G1(v0, v5, v10, v15, m[SG[r][8]], m[SG[r][9]]) G1(v1, v6, v11, v12, m[SG[r][10]], m[SG[r][11]])        \
    G1(v2, v7, v8, v13, m[SG[r][12]], m[SG[r][13]]) G1(v3, v4, v9, v14, m[SG[r][14]], m[SG[r][15]])
    // (sigma as compile-time constants: m[] stays in registers)
    constexpr uint8_t SG[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    ROUND1(0) ROUND1(1) ROUND1(2) ROUND1(3) ROUND1(4) ROUND1(5) ROUND1(6) ROUND1(7) ROUND1(8) ROUND1(9) ROUND1(10) ROUND1(11)
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// Everything the transcript absorbs in a proof is a whole number of 8-byte words (u64 lengths, 32-byte elements, 64-byte blocks): the
// buffer is 16 words.  absorb one word; a full buffer is compressed only when more input follows (as Blake2b512::update does).
struct Rng1 {
    uint64_t h[8];
    uint64_t *m, *mfin; // LDS, 16 words each
    uint64_t t;
    uint32_t fill; // words in m
    __device__ __forceinline__ void word(uint64_t w) {
        if (fill == 16) {
            t += 128;
            compress1(h, m, t, false);
            fill = 0;
        }
        m[fill] = w;
        fill += 1;
    }
    __device__ __forceinline__ void digest(uint64_t out[8]) const {
        uint64_t hh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hh[i] = h[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) mfin[i] = i < (int)fill ? m[i] : 0;
        compress1(hh, mfin, t + 8ULL * fill, true);
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = hh[i];
    }
    __device__ __forceinline__ uint64_t next_u64() { // rng.rs:51-55 over fill_bytes(8): digest, take 8 bytes, absorb the whole block
        uint64_t out[8];
        digest(out);
#pragma unroll
        for (int i = 0; i < 8; ++i) word(out[i]);
        return out[0];
    }
};

__device__ __constant__ uint64_t c_p[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
__device__ __forceinline__ bool geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > c_p[i]) return true;
        if (a[i] < c_p[i]) return false;
    }
    return true;
}

// round i's "message": D elements derived from the previous challenge (stands for the canonical form of the round polynomial)
__device__ __forceinline__ uint64_t msg_word(const uint64_t r[4], uint32_t e, uint32_t w) { return w == 3 ? ((r[3] >> 1) + e) : (r[w] ^ (0x9e3779b97f4a7c15ULL * (e + 1))); }

__global__ __launch_bounds__(64) void k_fs_one_lane(uint32_t rounds, uint32_t D, uint64_t *out, uint64_t *clocks) {
    __shared__ uint64_t m1[16], m1fin[16];
    if (threadIdx.x != 0) return;
    Rng1 g;
    g.m = m1;
    g.mfin = m1fin;
#pragma unroll
    for (int i = 0; i < 8; ++i) g.h[i] = c_iv[i];
    g.h[0] ^= 0x01010040ULL;
    g.t = 0;
    g.fill = 0;
    uint64_t r[4] = {1, 2, 3, 4};
    const uint64_t t0 = wall_clock64();
    for (uint32_t i = 0; i < rounds; ++i) {
        g.word((uint64_t)D); // CanonicalSerialize of Vec<F>: length, then the elements
        for (uint32_t e = 0; e < D; ++e)
            for (uint32_t w = 0; w < 4; ++w) g.word(msg_word(r, e, w));
        for (;;) { // F::rand
            uint64_t a[4];
            for (int w = 0; w < 4; ++w) a[w] = g.next_u64();
            a[3] &= 0x7fffffffffffffffULL;
            if (!geq_p(a)) {
                for (int w = 0; w < 4; ++w) r[w] = a[w];
                break;
            }
        }
    }
    const uint64_t t1 = wall_clock64();
    for (int w = 0; w < 4; ++w) out[w] = r[w];
    clocks[0] = t1 - t0;
}

// LDS traffic between the lanes of one wavefront: the hardware runs a wave's LDS instructions in order; this keeps the compiler from moving them
#define WAVE_SYNC()                                             \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

// ---- variant 1: four lanes, one column each ------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint64_t quad_rot(uint64_t x) { // lane j of every quad takes lane (j + k) & 3's value
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)x, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> 32), CTRL, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}
constexpr int kRot1 = 0x39, kRot2 = 0x4e, kRot3 = 0x93; // quad_perm [1,2,3,0] [2,3,0,1] [3,0,1,2]

// lds_m: the 16 message words; lane = threadIdx.x & 3.  h0/h1: this lane's two chaining words h[lane], h[4 + lane].
__device__ __forceinline__ void compress4(uint64_t &h0, uint64_t &h1, const uint64_t *lds_m, const uint8_t *lds_sigma, uint64_t t, bool last, int lane) {
    uint64_t a = h0, b = h1, c = c_iv[lane], d = c_iv[4 + lane];
    if (lane == 0) d ^= t;
    if (lane == 2 && last) d = ~d;
    // this lane's message words of the 12 rounds: all addresses are known up front, so every load is issued before the first G
    uint64_t mx[12][4];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        const uint8_t *s = lds_sigma + r * 16;
        mx[r][0] = lds_m[s[2 * lane]];
        mx[r][1] = lds_m[s[2 * lane + 1]];
        mx[r][2] = lds_m[s[8 + 2 * lane]];
        mx[r][3] = lds_m[s[8 + 2 * lane + 1]];
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        G1(a, b, c, d, mx[r][0], mx[r][1])
        b = quad_rot<kRot1>(b);
        c = quad_rot<kRot2>(c);
        d = quad_rot<kRot3>(d);
        G1(a, b, c, d, mx[r][2], mx[r][3])
        b = quad_rot<kRot3>(b);
        c = quad_rot<kRot2>(c);
        d = quad_rot<kRot1>(d);
    }
    h0 ^= a ^ c;
    h1 ^= b ^ d;
}

__global__ __launch_bounds__(64) void k_fs_four_lanes(uint32_t rounds, uint32_t D, uint64_t *out, uint64_t *clocks) {
    __shared__ uint64_t m[16];
    __shared__ uint64_t mfin[16];
    __shared__ uint64_t dig[8];
    __shared__ uint8_t sg[12 * 16];
    const int lane = threadIdx.x & 3;
    if (threadIdx.x >= 4) return; // one quad of one wavefront
    for (int i = lane; i < 12 * 16; i += 4) sg[i] = c_sigma[i / 16][i % 16];
    uint64_t h0 = c_iv[lane], h1 = c_iv[4 + lane];
    if (lane == 0) h0 ^= 0x01010040ULL;
    uint64_t t = 0;
    uint32_t fill = 0; // words in m (uniform over the quad)
    uint64_t r[4] = {1, 2, 3, 4};
    WAVE_SYNC();
    auto word = [&](uint64_t w) { // uniform call: lane 0 stores
        if (fill == 16) {
            t += 128;
            WAVE_SYNC();
            compress4(h0, h1, m, sg, t, false, lane);
            WAVE_SYNC();
            fill = 0;
        }
        if (lane == 0) m[fill] = w;
        fill += 1;
    };
    auto next_u64 = [&]() -> uint64_t {
        // finalise a clone: padded copy of the buffer, chaining words copied
        for (int i = lane; i < 16; i += 4) mfin[i] = i < (int)fill ? m[i] : 0;
        uint64_t g0 = h0, g1 = h1;
        WAVE_SYNC();
        compress4(g0, g1, mfin, sg, t + 8ULL * fill, true, lane);
        dig[lane] = g0;
        dig[4 + lane] = g1;
        WAVE_SYNC();
        uint64_t first = 0;
        for (int i = 0; i < 8; ++i) {
            const uint64_t w = dig[i];
            if (i == 0) first = w;
            word(w);
        }
        return first;
    };
    const uint64_t t0 = wall_clock64();
    for (uint32_t i = 0; i < rounds; ++i) {
        word((uint64_t)D);
        for (uint32_t e = 0; e < D; ++e)
            for (uint32_t w = 0; w < 4; ++w) word(msg_word(r, e, w));
        for (;;) {
            uint64_t a[4];
            for (int w = 0; w < 4; ++w) a[w] = next_u64();
            a[3] &= 0x7fffffffffffffffULL;
            if (!geq_p(a)) {
                for (int w = 0; w < 4; ++w) r[w] = a[w];
                break;
            }
        }
    }
    const uint64_t t1 = wall_clock64();
    if (lane == 0) {
        for (int w = 0; w < 4; ++w) out[w] = r[w];
        clocks[0] = t1 - t0;
    }
}

// ---- the host's transcript on the same chain -------------------------------------------------------------------------------
static void host_chain(uint32_t rounds, uint32_t D, uint64_t r[4], double *us_per_round) {
    sch::Blake2b512Rng rng;
    r[0] = 1; r[1] = 2; r[2] = 3; r[3] = 4;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < rounds; ++i) {
        uint64_t len = D;
        rng.feed_bytes(reinterpret_cast<const uint8_t *>(&len), 8);
        for (uint32_t e = 0; e < D; ++e) {
            uint64_t el[4];
            for (uint32_t w = 0; w < 4; ++w) el[w] = w == 3 ? ((r[3] >> 1) + e) : (r[w] ^ (0x9e3779b97f4a7c15ULL * (e + 1)));
            rng.feed_bytes(reinterpret_cast<const uint8_t *>(el), 32);
        }
        const sch::Fr c = rng.sample_fr();
        std::memcpy(r, c.l, 32);
    }
    *us_per_round = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
}

int main(int argc, char **argv) {
    const uint32_t rounds = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 256;
    uint64_t *d_out = nullptr, *d_clk = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&d_out), 64));
    CHECK(hipMalloc(reinterpret_cast<void **>(&d_clk), 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int bad = 0;
    std::printf("{\"rounds\": %u, \"per_round\": [", rounds);
    bool first = true;
    for (uint32_t D : {3u, 4u, 5u}) {
        uint64_t want[4];
        double host_us = 0;
        host_chain(rounds, D, want, &host_us); // (warm-up)
        host_chain(rounds, D, want, &host_us);
        for (int variant = 0; variant < 2; ++variant) {
            float best_ms = 1e30f;
            uint64_t clk = 0, got[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, nullptr));
                if (variant == 0) hipLaunchKernelGGL(k_fs_one_lane, dim3(1), dim3(64), 0, nullptr, rounds, D, d_out, d_clk);
                else hipLaunchKernelGGL(k_fs_four_lanes, dim3(1), dim3(64), 0, nullptr, rounds, D, d_out, d_clk);
                CHECK(hipEventRecord(e1, nullptr));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_ms) best_ms = ms;
                CHECK(hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
            }
            const bool equal = std::memcmp(got, want, 32) == 0;
            if (!equal) bad += 1;
            std::printf("%s\n  {\"D\": %u, \"variant\": \"%s\", \"device_us_events\": %.3f, \"device_us_clock\": %.3f, \"host_us\": %.3f, \"equal_to_host_transcript\": %s}", first ? "" : ",", D,
                        variant == 0 ? "one_lane" : "four_lanes_dpp", 1000.0 * best_ms / rounds, (double)clk * 0.01 / rounds, host_us, equal ? "true" : "false");
            first = false;
        }
    }
    std::printf("\n]}\n");
    return bad ? 1 : 0;
}
