// mem_bench.hip -- HBM streaming patterns of the bind kernel, isolated (no arithmetic):
//   A  lane reads its own 128 contiguous bytes (8 x dwordx4, lane stride 128 B)            [bind loads, reference layout]
//   B  the same bytes with every instruction contiguous across the wave (lane i: +16 i)    [ideal coalescing]
//   C  A + lane writes its own 64 contiguous bytes (4 x dwordx4, lane stride 64 B)         [bind loads + stores]
//   D  B + coalesced stores
//   E  lane reads 64 contiguous bytes (sum-only loads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void kA(const uint4 *src, uint4 *dst, uint64_t n_pairs, int store) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < n_pairs; b += stride) {
        const uint4 *p = src + 8 * b;
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
        if (store) {
            uint4 *q = dst + 4 * b;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = make_uint4(v[k].x ^ v[k + 4].x, v[k].y, v[k].z, v[k + 4].w);
        }
    }
    if (acc.x == 0x12345 && acc.y == 77) dst[0] = acc;
}
__global__ __launch_bounds__(256) void kB(const uint4 *src, uint4 *dst, uint64_t n_pairs, int store) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint64_t wave_stride = (uint64_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const uint64_t n_wt = n_pairs / 64; // wave tiles of 64 pairs
    for (uint64_t wt = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); wt < n_wt; wt += wave_stride) {
        const uint4 *p = src + wt * 512 + lane; // 8 KB per wave tile, instruction k reads 1 KB contiguous
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[64 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
        if (store) {
            uint4 *q = dst + wt * 256 + lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[64 * k] = make_uint4(v[k].x ^ v[k + 4].x, v[k].y, v[k].z, v[k + 4].w);
        }
    }
    if (acc.x == 0x12345 && acc.y == 77) dst[0] = acc;
}
__global__ __launch_bounds__(256) void kE(const uint4 *src, uint4 *dst, uint64_t n_pairs) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < n_pairs; b += stride) {
        const uint4 *p = src + 4 * b;
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
    }
    if (acc.x == 0x12345 && acc.y == 77) dst[0] = acc;
}
int main() {
    const uint64_t n_pairs = 1ull << 24; // 2 GiB read (128 B per pair), 1 GiB written
    uint4 *src, *dst;
    hipMalloc(&src, n_pairs * 128);
    hipMalloc(&dst, n_pairs * 64);
    hipMemset(src, 1, n_pairs * 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {1024, 2048, 4096}) {
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, 0, src, dst, n_pairs, 0);
                if (mode == 1) hipLaunchKernelGGL(kB, dim3(grid), dim3(256), 0, 0, src, dst, n_pairs, 0);
                if (mode == 2) hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, 0, src, dst, n_pairs, 1);
                if (mode == 3) hipLaunchKernelGGL(kB, dim3(grid), dim3(256), 0, 0, src, dst, n_pairs, 1);
                if (mode == 4) hipLaunchKernelGGL(kE, dim3(grid), dim3(256), 0, 0, src, dst, 2 * n_pairs);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)n_pairs * 128 + ((mode == 2 || mode == 3) ? (double)n_pairs * 64 : 0);
            const char *names[5] = {"A strided 128B/lane read", "B coalesced read", "C strided read+64B/lane write", "D coalesced read+write", "E strided 64B/lane read"};
            printf("grid %4d  %-32s %7.3f ms  %6.2f TB/s\n", grid, names[mode], best, bytes / best / 1e9);
        }
    }
    return 0;
}
