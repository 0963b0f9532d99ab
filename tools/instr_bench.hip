// instr_bench.hip -- per-instruction VALU issue cost on gfx950, relative to v_add_u32 (full rate).
// Build: hipcc --offload-arch=gfx950 -O3 tools/instr_bench.hip -o gpurun_out/instr_bench   (run on the GPU box)
// Every kernel runs REPS iterations of 32 copies of one instruction (independent destination registers where the
// encoding allows) on 256 CUs x 8 blocks x 256 threads, so the SIMDs are saturated and latency is hidden.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

#define KERNEL(name, body, ...)                                                                   \
    __global__ __launch_bounds__(256) void name(uint32_t reps, uint32_t *out) {                    \
        uint32_t a = threadIdx.x, b = blockIdx.x * 3 + 1, c = a ^ b, d = a + b;                    \
        uint64_t q = ((uint64_t)a << 32) | b, w = q * 3 + 1;                                       \
        double f = a * 1.5, g = b * 0.25, h = 1.0;                                                 \
        for (uint32_t i = 0; i < reps; ++i) {                                                      \
            asm volatile(REP32(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(q), "+v"(w), "+v"(f), "+v"(g), "+v"(h) : : __VA_ARGS__); \
        }                                                                                          \
        if ((a ^ b ^ c ^ d ^ (uint32_t)q ^ (uint32_t)w ^ (uint32_t)f ^ (uint32_t)g ^ (uint32_t)h) == 0x12345u) out[0] = a; \
    }

// operands: %0 a %1 b %2 c %3 d (32-bit) ; %4 q %5 w (64-bit pairs) ; %6 f %7 g %8 h (f64 pairs)
KERNEL(k_add_u32, "v_add_u32 %0, %1, %0\n", "memory")
KERNEL(k_mov, "v_mov_b32 %0, %1\n", "memory")
KERNEL(k_mad64_vcc, "v_mad_u64_u32 %4, vcc, %1, %2, %4\n", "vcc")
KERNEL(k_mad64_sgpr, "v_mad_u64_u32 %4, s[20:21], %1, %2, %4\n", "s20", "s21")
KERNEL(k_mad64_2acc, "v_mad_u64_u32 %4, s[20:21], %1, %2, %4\nv_mad_u64_u32 %5, s[22:23], %0, %3, %5\n", "s20", "s21", "s22", "s23")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %1, %2\n", "memory")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %1, %2\n", "memory")
KERNEL(k_addc_sgpr, "v_addc_co_u32 %0, s[20:21], 0, %0, s[22:23]\n", "s20", "s21")
KERNEL(k_add_co_vcc, "v_add_co_u32 %0, vcc, %1, %0\n", "vcc")
KERNEL(k_addco_nop_addc, "v_add_co_u32 %0, vcc, %1, %0\ns_nop 1\nv_addc_co_u32 %2, vcc, %3, %2, vcc\ns_nop 1\n", "vcc")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %4, %5, 0, %4\n", "memory")
KERNEL(k_fma_f64, "v_fma_f64 %8, %6, %7, %8\n", "memory")
KERNEL(k_mul_f64, "v_mul_f64 %8, %6, %7\n", "memory")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0\n", "memory")
KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %1, %2\n", "memory")
KERNEL(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %2\n", "memory")
KERNEL(k_add3, "v_add3_u32 %0, %1, %2, %0\n", "memory")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc\n", "memory")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %2, 7\n", "memory")
KERNEL(k_add_nop0, "v_add_u32 %0, %1, %0\ns_nop 0\n", "memory")
KERNEL(k_add_nop1, "v_add_u32 %0, %1, %0\ns_nop 1\n", "memory")
KERNEL(k_mad64_addc, "v_mad_u64_u32 %4, s[20:21], %1, %2, %4\nv_addc_co_u32 %0, s[22:23], 0, %0, s[22:23]\n", "s20", "s21", "s22", "s23")
KERNEL(k_lshrrev_b64, "v_lshrrev_b64 %4, 3, %5\n", "memory")
KERNEL(k_sub_u32, "v_sub_u32 %0, %1, %0\n", "memory")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %1, %0\n", "memory")
KERNEL(k_mad_u32_u16, "v_mad_u32_u16 %0, %1, %2, %0\n", "memory")
KERNEL(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0\n", "memory")
KERNEL(k_mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0\n", "memory")

KERNEL(k_madi64_dep, "v_mad_i64_i32 %4, s[20:21], %1, %2, %4\n", "s20", "s21") // one DEPENDENT chain (latency when few waves share a SIMD)
KERNEL(k_madi64_2dep, "v_mad_i64_i32 %4, s[20:21], %1, %2, %4\nv_mad_i64_i32 %5, s[22:23], %0, %3, %5\n", "s20", "s21", "s22", "s23")

template <typename K>
static double run(K kern, const char *name, int per_iter, double ref, int blocks_per_cu = 8) {
    uint32_t *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu, reps = 4000;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, 50u, d);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (uint32_t)reps, d);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs) * reps * 32 * per_iter
    const double winstr_per_simd = (double)blocks * 4 / 1024.0 * reps * 32.0 * per_iter;
    const double ns_per = best * 1e6 / winstr_per_simd;
    printf("%-22s %8.3f ms  %7.3f ns/wave-instr/SIMD  rel %.2f  (= %.2f cycles if v_add_u32 is 2)\n", name, best, ns_per,
           ref > 0 ? ns_per / ref : 1.0, ref > 0 ? 2.0 * ns_per / ref : 2.0);
    hipFree(d);
    return ns_per;
}

int main() {
    double ref = run(k_add_u32, "v_add_u32", 1, 0);
#define R(k, n) run(k, #k, n, ref)
    R(k_add_u32, 1); R(k_sub_u32, 1); R(k_mov, 1); R(k_add3, 1); R(k_cndmask, 1); R(k_alignbit, 1);
    R(k_mad64_vcc, 1); R(k_mad64_sgpr, 1); R(k_mad64_2acc, 2); R(k_mul_lo, 1); R(k_mul_hi, 1);
    R(k_addc_sgpr, 1); R(k_add_co_vcc, 1); R(k_addco_nop_addc, 2); R(k_mad64_addc, 2);
    R(k_lshl_add_u64, 1); R(k_lshrrev_b64, 1); R(k_fma_f64, 1); R(k_mul_f64, 1);
    R(k_mad_u32_u24, 1); R(k_mul_u32_u24, 1); R(k_mul_hi_u32_u24, 1); R(k_mad_i32_i24, 1); R(k_mad_u32_u16, 1); R(k_dot4_u32_u8, 1); R(k_pk_add_u16, 1);
    R(k_add_nop0, 1); R(k_add_nop1, 1);
    // dependent multiply-add chains at 1, 2, 3 wavefronts per SIMD (blocks of 256 threads = one wavefront on each of a CU's four SIMDs):
    // "ns per wave-instruction per SIMD" here is latency-limited; where it meets the saturated figure above, that many waves hide the latency
    for (int w = 1; w <= 4; ++w) {
        char nm[64];
        snprintf(nm, sizeof nm, "madi64 1 chain, %d w/SIMD", w);
        run(k_madi64_dep, nm, 1, ref, w);
        snprintf(nm, sizeof nm, "madi64 2 chains, %d w/SIMD", w);
        run(k_madi64_2dep, nm, 2, ref, w);
    }
    return 0;
}
