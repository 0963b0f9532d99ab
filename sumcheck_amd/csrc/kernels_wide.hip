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
#include "kernel_common.hpp"
#include "load_factor.hpp"

namespace scd {

// ---- extension weights (compile time) -------------------------------------------------------------------------------------------------
// a half of degree m (2..4) is known at the finite nodes F_m = {0, 1} (m = 2), {-1, 0, 1} (m = 3), {-1, 0, 1, 2} (m = 4) and by its leading
// coefficient ("inf"): v(x) = sum_{j in F_m} l_j(x) v(j) + N(x) v(inf), l_j the Lagrange basis over F_m, N(x) = prod_{j in F_m} (x - j).
constexpr int wide_first(int m) { return m >= 3 ? -1 : 0; }
constexpr int wide_last(int m) { return m >= 4 ? 2 : 1; }
constexpr long long wide_lagrange(int m, int j, int x) { // l_j(x): exact (consecutive integer nodes)
    long long num = 1, den = 1;
    for (int n = wide_first(m); n <= wide_last(m); ++n) {
        if (n == j) continue;
        num *= (x - n);
        den *= (j - n);
    }
    return num / den;
}
constexpr long long wide_lead(int m, int x) {
    long long w = 1;
    for (int n = wide_first(m); n <= wide_last(m); ++n) w *= (x - n);
    return w;
}
constexpr bool wide_in_base(int m, int j) { return j >= wide_first(m) && j <= wide_last(m); }
static_assert(wide_lagrange(4, -1, 3) == -1 && wide_lagrange(4, 0, 3) == 4 && wide_lagrange(4, 1, 3) == -6 && wide_lagrange(4, 2, 3) == 4 && wide_lead(4, 3) == 24, "degree 4 at node 3");
static_assert(wide_lagrange(2, 0, -1) == 2 && wide_lagrange(2, 1, -1) == -1 && wide_lead(2, -1) == 2, "q(-1) = 2 q(0) - q(1) + 2 q(inf)");
static_assert(wide_lagrange(3, -1, 2) == 1 && wide_lagrange(3, 0, 2) == -3 && wide_lagrange(3, 1, 2) == 3 && wide_lead(3, 2) == 6, "degree 3 at node 2");

// sum_i w_i v_i for five small integer weights (zero weights cost nothing), reduced to |value| < 2 p with limbs 0..7 in [0, 2^29):
// 64-bit columns (|w| < 2^8, |limb| < 2^30 + 8: no overflow), the quotient by p estimated from the top column in single precision
// (its error is far below one for |sum| < 2^12 p), q p taken off the columns, ONE carry chain.
__device__ __forceinline__ Fe fe_comb5(const Fe (&v)[5], const int w0, const int w1, const int wi, const int wm1, const int w2) {
    int64_t col[9];
#pragma unroll
    for (int l = 0; l < 9; ++l)
        col[l] = (int64_t)w0 * v[0].l[l] + (int64_t)w1 * v[1].l[l] + (int64_t)wi * v[2].l[l] + (int64_t)wm1 * v[3].l[l] + (int64_t)w2 * v[4].l[l];
    const int64_t top = col[8] + (col[7] >> 29);                                                   // ~ value / 2^232
    const int32_t q = (int32_t)((float)(int32_t)(top >> 4) * (16.0f / 7597479.4f));                // p / 2^232 = 0x73eda7.53...
#pragma unroll
    for (int l = 0; l < 9; ++l) col[l] -= (int64_t)q * (int64_t)fe_p_limb(l);
    Fe r;
    int64_t c = 0;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const int64_t t = col[l] + c;
        r.l[l] = (int32_t)(t & (int64_t)kFeMask);
        c = t >> 29; // arithmetic
    }
    r.l[8] = (int32_t)(col[8] + c);
    return r;
}

// a half's values: [0] at node 0, [1] at node 1, [2] its leading coefficient, [3] at node -1 (degree >= 3), [4] at node 2 (degree 4)
// value of the half (degree m in 2..4) at the product's node with index t (compile time): a base value, or the extension
template <int m, int t>
__device__ __forceinline__ Fe wide_value(const Fe (&v)[5]) {
    if constexpr (t <= m) {
        return v[t];
    } else {
        constexpr int x = node_value(t);
        static_assert(x != kNodeInf && !wide_in_base(m, x), "an extension node");
        constexpr int w0 = (int)wide_lagrange(m, 0, x), w1 = (int)wide_lagrange(m, 1, x), wi = (int)wide_lead(m, x);
        constexpr int wm1 = m >= 3 ? (int)wide_lagrange(m >= 3 ? m : 3, -1, x) : 0, w2 = m >= 4 ? (int)wide_lagrange(4, 2, x) : 0;
        return fe_comb5(v, w0, w1, wi, wm1, w2);
    }
}

// q = f g for two lines: its values at 0, 1 and its leading coefficient
template <bool kChain>
__device__ __forceinline__ void wide_quad(const Fe &l0, const Fe &h0, const Fe &l1, const Fe &h1, Fe &q0, Fe &q1, Fe &qi) {
    q0 = fe_mul<kChain>(l0, l1);
    q1 = fe_mul<kChain>(h0, h1);
    qi = fe_mul<kChain>(fe_sub(h0, l0), fe_sub(h1, l1));
}
// a quadratic at -1 and 2 from its values at 0, 1 and its leading coefficient (three lazy limb-wise additions and one carry pass each)
__device__ __forceinline__ Fe wide_quad_m1(const Fe &q0, const Fe &q1, const Fe &qi) { return fe_carry_pass(fe_sub(fe_add(fe_add(qi, qi), fe_add(q0, q0)), q1)); }
__device__ __forceinline__ Fe wide_quad_p2(const Fe &q0, const Fe &q1, const Fe &qi) { return fe_carry_pass(fe_sub(fe_add(fe_add(qi, qi), fe_add(q1, q1)), q0)); }

// the half made of factors F0 .. F0 + m - 1 of the product at pair b, multiplied out at its own nodes
template <int F0, int m, bool kChain>
__device__ __forceinline__ void wide_half(const Slot *S, const uint64_t b, const int32_t (&r)[kBindLds], Fe (&v)[5], Fe &lo1, Fe &hi1) {
    static_assert(m >= 1 && m <= 4, "a half has one to four factors");
    if constexpr (m == 1) {
        LoadFactor<F0, false, kChain>::run(S, b, r, lo1, hi1); // (a single factor: its line is evaluated node by node, fe_line)
        v[0] = lo1;
        v[1] = hi1;
        v[2] = fe_sub(hi1, lo1);
    } else if constexpr (m == 2) {
        Fe l0, h0, l1, h1;
        LoadFactor<F0, false, kChain>::run(S, b, r, l0, h0);
        LoadFactor<F0 + 1, false, kChain>::run(S, b, r, l1, h1);
        wide_quad<kChain>(l0, h0, l1, h1, v[0], v[1], v[2]);
    } else if constexpr (m == 3) {
        Fe q0, q1, qi;
        {
            Fe l0, h0, l1, h1;
            LoadFactor<F0, false, kChain>::run(S, b, r, l0, h0);
            LoadFactor<F0 + 1, false, kChain>::run(S, b, r, l1, h1);
            wide_quad<kChain>(l0, h0, l1, h1, q0, q1, qi);
        }
        Fe l2, h2;
        LoadFactor<F0 + 2, false, kChain>::run(S, b, r, l2, h2);
        const Fe s2 = fe_sub(h2, l2);
        v[0] = fe_mul<kChain>(l2, q0);
        v[1] = fe_mul<kChain>(h2, q1);
        v[2] = fe_mul<kChain>(s2, qi);
        v[3] = fe_mul<kChain>(fe_carry_pass(fe_sub(l2, s2)), wide_quad_m1(q0, q1, qi)); // the line at -1: 2 lo - hi
    } else {
        Fe a0, a1, ai, b0, b1, bi;
        {
            Fe l0, h0, l1, h1;
            LoadFactor<F0, false, kChain>::run(S, b, r, l0, h0);
            LoadFactor<F0 + 1, false, kChain>::run(S, b, r, l1, h1);
            wide_quad<kChain>(l0, h0, l1, h1, a0, a1, ai);
        }
        {
            Fe l2, h2, l3, h3;
            LoadFactor<F0 + 2, false, kChain>::run(S, b, r, l2, h2);
            LoadFactor<F0 + 3, false, kChain>::run(S, b, r, l3, h3);
            wide_quad<kChain>(l2, h2, l3, h3, b0, b1, bi);
        }
        v[0] = fe_mul<kChain>(a0, b0);
        v[1] = fe_mul<kChain>(a1, b1);
        v[2] = fe_mul<kChain>(ai, bi);
        v[3] = fe_mul<kChain>(wide_quad_m1(a0, a1, ai), wide_quad_m1(b0, b1, bi));
        v[4] = fe_mul<kChain>(wide_quad_p2(a0, a1, ai), wide_quad_p2(b0, b1, bi));
    }
}

template <int M, int t>
struct WideNodes { // nodes t .. M of one pair: the halves' values meet, the product joins node t's running sum
    template <typename Acc>
    static __device__ __forceinline__ void run(const Fe (&A)[5], const Fe (&B)[5], const Fe &lo1, const Fe &hi1, const Acc &accumulate) {
        constexpr int mb = M - 4;
        const Fe a = wide_value<4, t>(A);
        Fe bv;
        if constexpr (mb == 1) {
            // a single factor: its line at node x is (1 - x) lo + x hi -- as a REDUCED combination, not fe_line's lazy one: entries of
            // the internal-format tables are lazy sums of up to (round + 1) p, and nine times that leaves the products' limb bounds
            if constexpr (t <= 2) bv = B[t];
            else bv = fe_comb5(B, 1 - node_value(t), node_value(t), 0, 0, 0);
        }
        else bv = wide_value<mb < 2 ? 2 : mb, t>(B);
        accumulate(t, fe_mul<kChainDefault>(bv, a));
        if constexpr (t < M) WideNodes<M, t + 1>::run(A, B, lo1, hi1, accumulate);
    }
};

constexpr int kWideBlock = 256;
template <int M>
__global__ __launch_bounds__(kWideBlock) void k_prod_tree_wide(const ProdArgs P, const BindConst r, const uint64_t n_pairs, uint4 *__restrict__ partials) {
    static_assert(M >= 5 && M <= 8, "five to eight multiplicands");
    __shared__ uint32_t sm[kWideBlock / 64][8];
    __shared__ int32_t rt[kBindLds];
    extern __shared__ int32_t wide_lacc[]; // the M + 1 running sums, limb-planar, one column per thread (private: no barrier)
    bind_consts_to_lds(r, rt);
    __syncthreads();
    int32_t *my = wide_lacc + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 9 * (M + 1); ++i) my[i * kWideBlock] = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kWideBlock;
    uint32_t iter = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * kWideBlock + threadIdx.x; b < n_pairs; b += stride, ++iter) {
        auto accumulate = [&](const int t, const Fe &v) {
            Fe acc;
#pragma unroll
            for (int l = 0; l < 9; ++l) acc.l[l] = my[(9 * t + l) * kWideBlock];
            acc = fe_carry_pass(fe_add(acc, v));
            if ((iter & 31u) == 31u) acc = fe_from_fr(fe_to_fr(acc)); // keep the top limb far from 2^31 on long grid-stride loops
#pragma unroll
            for (int l = 0; l < 9; ++l) my[(9 * t + l) * kWideBlock] = acc.l[l];
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
        for (int l = 0; l < 9; ++l) a.l[l] = my[(9 * t + l) * kWideBlock];
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
static hipError_t launch_wide_t(const ProdArgs &args, const BindConst &rc, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream) {
    const size_t lds = (size_t)9 * (M + 1) * kWideBlock * 4;
    static bool attr_set = false; // (more dynamic LDS than the default limit of a launch: 55-83 KB of running sums)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_prod_tree_wide<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_prod_tree_wide<M>, dim3(grid), dim3(kWideBlock), lds, stream, args, rc, n_pairs, (uint4 *)d_partials);
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
