// wide_tree.hpp -- device helpers of the product trees with node extension (kernels_wide.hip: five to eight multiplicands;
// kernels_wide16.hip: nine to twelve): a half's static product tree, the integer Lagrange extension of its values to further nodes.
#pragma once
#include "kernel_common.hpp"
#include "load_factor.hpp"

namespace scd {

// ---- extension weights (compile time) -------------------------------------------------------------------------------------------------
// a half of degree m (2..4) is known at the finite nodes F_m = {0, 1} (m = 2), {-1, 0, 1} (m = 3), {-1, 0, 1, 2} (m = 4) and by its leading
// coefficient ("inf"): v(x) = sum_{j in F_m} l_j(x) v(j) + N(x) v(inf), l_j the Lagrange basis over F_m, N(x) = prod_{j in F_m} (x - j).
// (in general: the m finite nodes among the node indices 0 .. m are the consecutive integers -((m - 1) / 2) .. m / 2)
constexpr int wide_first(int m) { return -((m - 1) / 2); }
constexpr int wide_last(int m) { return m / 2; }
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

// ---- the same for any degree up to eight (kernels_wide16.hip) ---------------------------------------------------------------------------
// A polynomial of degree m (1..8) is held as its values at the node INDICES 0 .. max(m, 2) (kernels.h: node_value; index 2 is the
// leading coefficient).  Its value at any further node index t is sum_s w[s] v[s] with integer weights -- |w| < 2^23 for m = 8 at the
// nodes of a product of sixteen -- so the columns stay 64-bit sums of multiply-adds; the quotient by p is estimated in double precision
// (|sum| < 2^28 p: the single-precision estimate of fe_comb5 is not enough here), q p taken off the columns, one carry chain.
struct WideWeights {
    int w[9];
};
constexpr WideWeights wide_weights(int m, int t) {
    WideWeights W{};
    const int x = node_value(t);
    for (int s = 0; s <= 8; ++s) {
        W.w[s] = 0;
        if (s > (m > 2 ? m : 2)) continue;
        if (s == 2) {
            W.w[s] = (int)wide_lead(m, x);
            continue;
        }
        const int j = node_value(s);
        if (wide_in_base(m, j)) W.w[s] = (int)wide_lagrange(m, j, x);
    }
    return W;
}
static_assert(wide_first(8) == -3 && wide_last(8) == 4 && wide_first(5) == -2 && wide_last(5) == 2 && wide_first(1) == 0 && wide_last(1) == 0, "finite nodes of a degree");
static_assert(wide_weights(1, 4).w[0] == 1 && wide_weights(1, 4).w[1] == 0 && wide_weights(1, 4).w[2] == 2, "a line at node 2: lo + 2 slope");
static_assert(wide_weights(8, 9).w[2] == 40320 && wide_weights(8, 9).w[8] == -1 && wide_weights(8, 16).w[2] == 6652800, "degree 8 at node -4: lead weight 8!, the value at 4 enters with -1; at node 8: 11!/3!");

template <int m, int t>
__device__ __forceinline__ Fe wide_ext(const Fe (&v)[9]) {
    constexpr WideWeights W = wide_weights(m, t);
    int64_t col[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) {
        col[l] = 0;
#pragma unroll
        for (int s = 0; s < 9; ++s)
            if (W.w[s] != 0) col[l] += (int64_t)W.w[s] * (int64_t)v[s].l[l];
    }
    const int64_t top = col[8] + (col[7] >> 29);                    // ~ value / 2^232
    const int32_t q = (int32_t)((double)top * (1.0 / 7597479.3249)); // p / 2^232 = 0x73eda7.53...
#pragma unroll
    for (int l = 0; l < 9; ++l) col[l] -= (int64_t)q * (int64_t)fe_p_limb(l);
    Fe r;
    int64_t c = 0;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const int64_t tt = col[l] + c;
        r.l[l] = (int32_t)(tt & (int64_t)kFeMask);
        c = tt >> 29; // arithmetic
    }
    r.l[8] = (int32_t)(col[8] + c);
    return r;
}

// the product of the m factors F0 .. F0 + m - 1 (m = 1..8) at pair b, at ITS OWN node indices 0 .. max(m, 2)
template <int F0, int m, bool kChain>
__device__ __forceinline__ void wide_values(const Slot *S, const uint64_t b, const int32_t (&r)[kBindLds], Fe (&v)[9]) {
    static_assert(m >= 1 && m <= 8, "one to eight factors");
    Fe lo1, hi1;
    if constexpr (m <= 4) {
        Fe h[5];
        wide_half<F0, m, kChain>(S, b, r, h, lo1, hi1);
#pragma unroll
        for (int s = 0; s <= (m > 2 ? m : 2); ++s) v[s] = h[s];
    } else {
        Fe A[5], B[5];
        wide_half<F0, 4, kChain>(S, b, r, A, lo1, hi1);
        wide_half<F0 + 4, m - 4, kChain>(S, b, r, B, lo1, hi1);
        WideNodes<m, 0>::run(A, B, lo1, hi1, [&](const int t, const Fe &x) { v[t] = x; });
    }
}

} // namespace scd
