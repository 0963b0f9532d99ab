// mem_policy_bench.hip -- the big rounds' two streaming patterns (no arithmetic) under every cache-policy modifier of gfx950's global loads / stores:
//   R2  lane reads its own 128 contiguous bytes (the caller's canonical tables, cached loads) and the wave writes contiguous kilobytes (policy on the stores)
//   R3  contiguous kilobyte loads (policy) and contiguous kilobyte stores (policy), two bytes read per byte written
// hipcc --offload-arch=gfx950 -O3 tools/mem_policy_bench.hip -o /tmp/mem_policy_bench && /tmp/mem_policy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LOAD_FN(name, mod)                                                                                         \
    __device__ __forceinline__ u32x4 name(const u32x4 *p) {                                                        \
        u32x4 v;                                                                                                   \
        asm volatile("global_load_dwordx4 %0, %1, off " mod : "=v"(v) : "v"(p) : "memory");                        \
        return v;                                                                                                  \
    }
#define STORE_FN(name, mod)                                                                                        \
    __device__ __forceinline__ void name(u32x4 *p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off " mod ::"v"(p), "v"(v) : "memory"); }
LOAD_FN(ld0, "")
LOAD_FN(ld1, "nt")
LOAD_FN(ld2, "sc0")
LOAD_FN(ld3, "sc1")
LOAD_FN(ld4, "sc0 sc1")
LOAD_FN(ld5, "sc0 nt")
LOAD_FN(ld6, "sc1 nt")
LOAD_FN(ld7, "sc0 sc1 nt")
STORE_FN(st0, "")
STORE_FN(st1, "nt")
STORE_FN(st2, "sc0")
STORE_FN(st3, "sc1")
STORE_FN(st4, "sc0 sc1")
STORE_FN(st5, "sc0 nt")
STORE_FN(st6, "sc1 nt")
STORE_FN(st7, "sc0 sc1 nt")
template <int L> __device__ __forceinline__ u32x4 ld(const u32x4 *p) {
    if constexpr (L == 0) return ld0(p); else if constexpr (L == 1) return ld1(p); else if constexpr (L == 2) return ld2(p); else if constexpr (L == 3) return ld3(p);
    else if constexpr (L == 4) return ld4(p); else if constexpr (L == 5) return ld5(p); else if constexpr (L == 6) return ld6(p); else return ld7(p);
}
template <int S> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) {
    if constexpr (S == 0) st0(p, v); else if constexpr (S == 1) st1(p, v); else if constexpr (S == 2) st2(p, v); else if constexpr (S == 3) st3(p, v);
    else if constexpr (S == 4) st4(p, v); else if constexpr (S == 5) st5(p, v); else if constexpr (S == 6) st6(p, v); else st7(p, v);
}
// R2: strided cached loads (plain C++), coalesced stores with policy S
template <int S> __global__ __launch_bounds__(256) void kR2(const u32x4 *src, u32x4 *dst, uint64_t n_pairs) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < n_pairs; b += stride) {
        const u32x4 *p = src + 8 * b;
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[k];
        u32x4 *q = dst + (b - lane) * 4 + lane; // the wave's 64 pairs x 64 B = 4 KB, instruction k covers 1 KB
#pragma unroll
        for (int k = 0; k < 4; ++k) st<S>(q + 64 * k, v[k] ^ v[k + 4]);
    }
}
// R3: coalesced loads with policy L, coalesced stores with policy S
template <int L, int S> __global__ __launch_bounds__(256) void kR3(const u32x4 *src, u32x4 *dst, uint64_t n_pairs) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < n_pairs; b += stride) {
        const u32x4 *p = src + (b - lane) * 8 + lane;
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ld<L>(p + 64 * k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4 *q = dst + (b - lane) * 4 + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) st<S>(q + 64 * k, v[k] ^ v[k + 4]);
    }
}
static const char *kMods[8] = {"default", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
template <typename F> static float best_of(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
template <int S> static void run_r2(const u32x4 *src, u32x4 *dst, uint64_t n) {
    const float ms = best_of([&] { hipLaunchKernelGGL(kR2<S>, dim3(3072), dim3(256), 0, 0, src, dst, n); });
    printf("R2  loads cached/strided  stores %-11s %7.3f ms  %5.2f TB/s\n", kMods[S], ms, (double)n * 192 / ms / 1e9);
}
template <int L, int S> static void run_r3(const u32x4 *src, u32x4 *dst, uint64_t n) {
    const float ms = best_of([&] { hipLaunchKernelGGL((kR3<L, S>), dim3(3072), dim3(256), 0, 0, src, dst, n); });
    printf("R3  loads %-11s stores %-11s %7.3f ms  %5.2f TB/s\n", kMods[L], kMods[S], ms, (double)n * 192 / ms / 1e9);
}
int main() {
    const uint64_t n = 1ull << 24; // 2 GiB read, 1 GiB written
    u32x4 *src, *dst;
    hipMalloc(&src, n * 128);
    hipMalloc(&dst, n * 64);
    hipMemset(src, 1, n * 128);
    run_r2<0>(src, dst, n); run_r2<1>(src, dst, n); run_r2<2>(src, dst, n); run_r2<3>(src, dst, n);
    run_r2<4>(src, dst, n); run_r2<5>(src, dst, n); run_r2<6>(src, dst, n); run_r2<7>(src, dst, n);
    run_r3<0, 0>(src, dst, n); run_r3<0, 1>(src, dst, n); run_r3<1, 0>(src, dst, n); run_r3<1, 1>(src, dst, n);
    run_r3<2, 1>(src, dst, n); run_r3<3, 1>(src, dst, n); run_r3<4, 1>(src, dst, n); run_r3<5, 1>(src, dst, n); run_r3<6, 1>(src, dst, n); run_r3<7, 1>(src, dst, n);
    run_r3<1, 2>(src, dst, n); run_r3<1, 3>(src, dst, n); run_r3<1, 4>(src, dst, n); run_r3<1, 5>(src, dst, n); run_r3<1, 6>(src, dst, n); run_r3<1, 7>(src, dst, n);
    run_r3<6, 6>(src, dst, n); run_r3<5, 5>(src, dst, n); run_r3<7, 7>(src, dst, n); run_r3<3, 3>(src, dst, n);
    return 0;
}
