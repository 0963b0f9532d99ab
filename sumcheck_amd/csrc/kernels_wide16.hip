// kernels_wide16.hip -- big rounds of a product of NINE TO TWELVE multiplicands (k_prod_tree_wide16<M>): everything the reference's own
// tests build (ml_sumcheck/test.rs:125,191: 4..12 multiplicands per product).
//
// One level above kernels_wide.hip: the product is split into the product of its first eight factors -- itself two four-factor trees
// extended and multiplied at nine nodes (wide_tree.hpp: wide_values) -- and the product of the remaining one to four; both are extended
// to the product's further nodes as integer combinations of their own node values (wide_ext: weights below 2^23) and meet in ONE
// carry-free product per node.  M = 12: 31 + 11 + 13 = 55 products and twelve extensions per pair, where the node-by-node kernel
// (k_sum_generic) spends 13 x 24 saturated products behind a separate bind pass.
// The tables are this round's (bound by the bind pass that precedes a launch with products beyond kMaxFusedM; canonical reference
// layout); one slot per FACTOR.  The sums leave the kernel in the node-by-node kernel's form -- plain Montgomery sums, same partial
// layout -- so the finalize step and the small rounds see no difference: each block's sums are multiplied by 2^(5(M-1)) (`comp`) on
// the way out, undoing the 2^-5 every carry-free product carries.
#include "wide_tree.hpp"

namespace scd {

constexpr int kWide16Block = 128; // (the M + 1 running sums live in LDS, 36 (M + 1) bytes a thread: 60 KB a block for M = 12)

template <int M, int t>
struct Wide16Nodes {
    template <typename Acc>
    static __device__ __forceinline__ void run(const Fe (&A)[9], const Fe (&B)[9], const Acc &accumulate) {
        constexpr int mb = M - 8;
        Fe a, bv;
        if constexpr (t <= 8) a = A[t];
        else a = wide_ext<8, t>(A);
        if constexpr (t <= (mb > 2 ? mb : 2)) bv = B[t];
        else bv = wide_ext<mb, t>(B);
        accumulate(t, fe_mul<kChainDefault>(a, bv));
        if constexpr (t < M) Wide16Nodes<M, t + 1>::run(A, B, accumulate);
    }
};

template <int M>
__global__ __launch_bounds__(kWide16Block) void k_prod_tree_wide16(const WideArgs16 P, const FrHost comp_h, const uint64_t n_pairs, uint4 *__restrict__ partials) {
    static_assert(M >= 9 && M <= 12, "nine to twelve multiplicands");
    __shared__ int32_t rt[kBindLds];        // (the factor loader's signature: every slot is mode 0, nothing is bound here)
    extern __shared__ int32_t wide_lacc[]; // the M + 1 running sums, limb-planar, one column per thread (private: no barrier)
    int32_t *my = wide_lacc + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 9 * (M + 1); ++i) my[i * kWide16Block] = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kWide16Block;
    uint32_t iter = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * kWide16Block + threadIdx.x; b < n_pairs; b += stride, ++iter) {
        auto accumulate = [&](const int t, const Fe &v) {
            Fe acc;
#pragma unroll
            for (int l = 0; l < 9; ++l) acc.l[l] = my[(9 * t + l) * kWide16Block];
            acc = fe_carry_pass(fe_add(acc, v));
            if ((iter & 31u) == 31u) acc = fe_from_fr(fe_to_fr(acc)); // keep the top limb far from 2^31 on long grid-stride loops
#pragma unroll
            for (int l = 0; l < 9; ++l) my[(9 * t + l) * kWide16Block] = acc.l[l];
        };
        Fe A[9], B[9];
        wide_values<0, 8, kChainDefault>(P.slot, b, rt, A);
        wide_values<8, M - 8, kChainDefault>(P.slot, b, rt, B);
        Wide16Nodes<M, 0>::run(A, B, accumulate);
    }
    // block sums of the M + 1 nodes together (as kernels_wide.hip), then the scale comes off
    Fr sv[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) {
        Fe a;
#pragma unroll
        for (int l = 0; l < 9; ++l) a.l[l] = my[(9 * t + l) * kWide16Block];
        sv[t] = fe_to_fr(a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int t = 0; t <= M; ++t) sv[t] = fr_add(sv[t], fr_shfl_down(sv[t], off));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *x = reinterpret_cast<uint32_t *>(wide_lacc); // [wave][node][8]
    __syncthreads();                                        // every thread has read its running sums
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t <= M; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[(wave * (M + 1) + t) * 8 + i] = sv[t].v[i];
    }
    __syncthreads();
    if (threadIdx.x <= (uint32_t)M) {
        const int t = threadIdx.x;
        Fr acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.v[i] = x[t * 8 + i];
        for (int w = 1; w < kWide16Block / 64; ++w) {
            Fr o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = x[(w * (M + 1) + t) * 8 + i];
            acc = fr_add(acc, o);
        }
        fr_store(partials + 2 * ((uint64_t)t * gridDim.x + blockIdx.x), fr_mul(acc, fr_from_host(comp_h)));
    }
}

template <int M>
static hipError_t launch_wide16_t(const WideArgs16 &args, const FrHost &comp, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    const size_t lds = (size_t)9 * (M + 1) * kWide16Block * 4;
    static bool attr_set[64] = {}; // (dynamic LDS near the default limit of a launch); per device
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_prod_tree_wide16<M>), (int)lds, attr_set); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_prod_tree_wide16<M>, dim3(grid), dim3(kWide16Block), lds, stream, args, comp, n_pairs, (uint4 *)d_partials);
    return hipGetLastError();
}

hipError_t launch_prod_tree_wide16(int M, const WideArgs16 &args, const FrHost &comp, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    switch (M) {
    case 9: return launch_wide16_t<9>(args, comp, n_pairs, d_partials, grid, stream);
    case 10: return launch_wide16_t<10>(args, comp, n_pairs, d_partials, grid, stream);
    case 11: return launch_wide16_t<11>(args, comp, n_pairs, d_partials, grid, stream);
    case 12: return launch_wide16_t<12>(args, comp, n_pairs, d_partials, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace scd
