// kernels_wide.hip -- big rounds of a product of FIVE TO EIGHT multiplicands as a product tree with node extension (k_prod_tree_wide<M>).
//
// The round polynomial of a product of M multilinears has degree M: M + 1 node sums (nodes 0, 1, inf, -1, 2, -2, 3, -3, 4 in the kernels'
// order, kernels.h: node_value).  Node by node (k_prod_round_fe<M>) that is (M + 1)(M - 1) Montgomery products per pair -- 63 for eight
// multiplicands, behind 336 bytes of scratch per lane.  Here the product is split into the half of its first four factors and the
// half of the rest (1..4 factors); each half is multiplied out by the static tree of the four-multiplicand kernels at ITS OWN degree + 1
// nodes (0 / 3 / 7 / 11 products), EXTENDED to the product's remaining nodes without a multiplication -- a polynomial of degree m known at
// m consecutive integers and by its leading coefficient is, at any other integer, an INTEGER combination of those values (Lagrange over
// consecutive nodes; weights below 2^8 for everything used here): nine small-constant multiply-adds per limb column, one quotient
// estimate, one carry chain -- and the halves meet in ONE product per node: M = 8: 11 + 11 + 9 = 31 products and eight extensions
// (each about half a product) instead of 63; M = 5: 11 + 0 + 6 and one extension instead of 24.
// Same arguments (one slot per factor: load_factor.hpp), partial layout and 2^(-5(M-1)) scaling as k_prod_tree<M> / k_prod_round_fe<M>.
#include "wide_tree.hpp"

namespace scd {

constexpr int kWideBlock = 256;
// kPairSums: adjacent lanes add their node products (one DPP add per limb) and the even lane alone keeps the running sum -- half the LDS, so
// that two blocks of an eight-multiplicand product fit a CU (2 x 41.5 KB instead of 83 KB each)
template <int M, bool kPairSums = false>
__device__ __forceinline__ void prod_tree_wide_body(const ProdArgs &P, const BindConst &r, const uint64_t n_pairs, uint4 *__restrict__ partials) {
    static_assert(M >= 5 && M <= 8, "five to eight multiplicands");
    constexpr int kCols = kPairSums ? kWideBlock / 2 : kWideBlock;
    __shared__ uint32_t sm[kWideBlock / 64][8];
    __shared__ int32_t rt[kBindLds];
    extern __shared__ int32_t wide_lacc[]; // the M + 1 running sums, limb-planar, one column per thread or lane pair (private: no barrier)
    bind_consts_to_lds(r, rt);
    __syncthreads();
    int32_t *my = wide_lacc + (kPairSums ? threadIdx.x >> 1 : threadIdx.x);
    const bool keeper = !kPairSums || (threadIdx.x & 1u) == 0;
    if (keeper) {
#pragma unroll
        for (int i = 0; i < 9 * (M + 1); ++i) my[i * kCols] = 0;
    }
    const uint64_t stride = (uint64_t)gridDim.x * kWideBlock;
    uint32_t iter = 0;
    // (n_pairs and the stride are even: the two lanes of a pair leave the loop together)
    for (uint64_t b = (uint64_t)blockIdx.x * kWideBlock + threadIdx.x; b < n_pairs; b += stride, ++iter) {
        auto accumulate = [&](const int t, const Fe &v_in) {
            Fe v = v_in;
            if constexpr (kPairSums) {
#pragma unroll
                for (int l = 0; l < 9; ++l) v.l[l] += __builtin_amdgcn_mov_dpp(v_in.l[l], 0xB1, 0xF, 0xF, true); // quad_perm [1, 0, 3, 2]: the neighbour's limb
            }
            if (keeper) {
                Fe acc;
#pragma unroll
                for (int l = 0; l < 9; ++l) acc.l[l] = my[(9 * t + l) * kCols];
                acc = fe_carry_pass(fe_add(acc, v));
                if ((iter & (kPairSums ? 15u : 31u)) == (kPairSums ? 15u : 31u)) acc = fe_from_fr(fe_to_fr(acc)); // keep the top limb far from 2^31 on long grid-stride loops
#pragma unroll
                for (int l = 0; l < 9; ++l) my[(9 * t + l) * kCols] = acc.l[l];
            }
        };
        Fe A[5], B[5], lo1, hi1;
        wide_half<0, 4, kChainDefault>(P.slot, b, rt, A, lo1, hi1);
        wide_half<4, M - 4, kChainDefault>(P.slot, b, rt, B, lo1, hi1);
        WideNodes<M, 0>::run(A, B, lo1, hi1, accumulate);
    }
    // Block sums of the M + 1 nodes TOGETHER (as the four-multiplicand kernels do): the canonical conversions and the six shuffle steps of
    // the nodes are independent chains that the scheduler interleaves; the wavefronts' sums cross through LDS behind ONE pair of barriers
    // (the running sums' LDS is free by then) and threads 0..M each finish one node.  Node after node this epilogue was 36 us of a
    // 48 us launch in the short rounds.
    (void)sm;
    Fr sv[M + 1];
#pragma unroll
    for (int t = 0; t <= M; ++t) {
        Fe a;
#pragma unroll
        for (int l = 0; l < 9; ++l) a.l[l] = keeper ? my[(9 * t + l) * kCols] : 0;
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
        for (int w = 1; w < kWideBlock / 64; ++w) {
            Fr o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = x[(w * (M + 1) + t) * 8 + i];
            acc = fr_add(acc, o);
        }
        fr_store(partials + 2 * ((uint64_t)t * gridDim.x + blockIdx.x), acc);
    }
}

template <int M>
__global__ __launch_bounds__(kWideBlock) void k_prod_tree_wide(const ProdArgs P, const BindConst r, const uint64_t n_pairs, uint4 *__restrict__ partials) {
    prod_tree_wide_body<M>(P, r, n_pairs, partials);
}
// Seven multiplicands: 281 registers as the compiler schedules it freely -- one wavefront per SIMD -- although two blocks' running sums
// (74 KB each) fit a CU's LDS.  Held to 256 registers (18 of them spilled) two blocks are resident: the 4 / 5 / 6 / 7 / 8 mix at nv = 20
// 2.98 -> 2.88 ms on the same box (profiles/r6i_wide_occupancy_ab.txt).  Eight multiplicands: 83 KB of running sums allow one block per CU
// whatever the registers (held to 256 alone, the 60 spilled registers cost 4 %: 4.52 -> 4.69 ms for five products of eight); with the
// sums of lane PAIRS (kPairSums: 41.5 KB a block) two blocks fit.
template <int M>
__global__ __launch_bounds__(kWideBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_prod_tree_wide_two_blocks(const ProdArgs P, const BindConst r, const uint64_t n_pairs,
                                                                                                                uint4 *__restrict__ partials) {
    prod_tree_wide_body<M, (M == 8)>(P, r, n_pairs, partials);
}

template <int M>
static hipError_t launch_wide_t(const ProdArgs &args, const BindConst &rc, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    static bool attr_set[64] = {}, attr_set_two[64] = {}; // (more dynamic LDS than the default limit of a launch: 41-83 KB of running sums); per device and kernel
    // (the pair sums assume that the two lanes of a pair leave the grid-stride loop together: an even number of pairs -- every big round has a
    // power of two of at least 2^15; anything else takes the one-block kernel)
    if (M >= 7 && (M == 7 || (n_pairs & 1) == 0)) {
        const size_t lds = (size_t)9 * (M + 1) * (M == 8 ? kWideBlock / 2 : kWideBlock) * 4;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_prod_tree_wide_two_blocks<M>), (int)lds, attr_set_two); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_prod_tree_wide_two_blocks<M>, dim3(grid), dim3(kWideBlock), lds, stream, args, rc, n_pairs, (uint4 *)d_partials);
    } else {
        const size_t lds = (size_t)9 * (M + 1) * kWideBlock * 4;
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_prod_tree_wide<M>), (int)lds, attr_set); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_prod_tree_wide<M>, dim3(grid), dim3(kWideBlock), lds, stream, args, rc, n_pairs, (uint4 *)d_partials);
    }
    return hipGetLastError();
}

hipError_t launch_prod_tree_wide(int M, const ProdArgs &args, const BindConst &rc, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    switch (M) {
    case 5: return launch_wide_t<5>(args, rc, n_pairs, d_partials, grid, stream);
    case 6: return launch_wide_t<6>(args, rc, n_pairs, d_partials, grid, stream);
    case 7: return launch_wide_t<7>(args, rc, n_pairs, d_partials, grid, stream);
    case 8: return launch_wide_t<8>(args, rc, n_pairs, d_partials, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace scd
