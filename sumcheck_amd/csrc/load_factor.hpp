// load_factor.hpp -- one factor of a product at one pair, as the fused bind + sum kernels take it (kernels_big.hip: product tree for
// up to four multiplicands; kernels_wide.hip: five to eight).  A slot lists ONE factor (a table that occurs twice in a product is listed
// twice).  Slot modes: 0 read this round's table; 1 bind the previous table, store, use; 3 bind without storing (a repeated factor, or a
// table that an earlier product of the same launch stores).
#pragma once
#include "kernel_common.hpp"

namespace scd {

// factor F of the product at pair b: its line's two end points (lo, hi), binding / storing as the slot's mode says
// kR1: round 1 of a proof -- every factor is read from the caller's canonical table, nothing is bound (k_round1_tree)
template <int F, bool kR1 = false, bool kChain = kChainDefault>
struct LoadFactor {
    static __device__ __forceinline__ void run(const Slot *S, const uint64_t b, const int32_t (&r)[kBindLds], Fe &lo_out, Fe &hi_out) {
        const Slot &sl = S[F];
        if constexpr (kR1) {
            const uint4 *p = sl.src + 4 * b;
            lo_out = fe_from_fr(fr_load(p));
            hi_out = fe_from_fr(fr_load(p + 2));
            return;
        }
        const uint32_t mode = sl.mode;
        const int32_t *stop = sl.src_top; // non-null: the source table is in the internal F29 format
        if (mode == 0) {
            const uint4 *p = sl.src + 4 * b;
            if (stop) {
                const int2 t = *reinterpret_cast<const int2 *>(stop + 2 * b);
                lo_out = fe_load_f29(sl.src, 2 * b, t.x);
                hi_out = fe_load_f29(sl.src, 2 * b + 1, t.y);
            } else {
                lo_out = fe_from_fr(fr_load(p));
                hi_out = fe_from_fr(fr_load(p + 2));
            }
        } else {
            const uint4 *p = sl.src + 8 * b; // entries 4b..4b+3 of the previous table: 128 contiguous bytes
            Fe e0, e1, e2, e3;
            if (stop) {
                const int4 t = *reinterpret_cast<const int4 *>(stop + 4 * b);
                const uint4 *m = sl.src;
                e0 = fe_load_f29(m, 4 * b, t.x); e1 = fe_load_f29(m, 4 * b + 1, t.y);
                e2 = fe_load_f29(m, 4 * b + 2, t.z); e3 = fe_load_f29(m, 4 * b + 3, t.w);
            } else {
                e0 = fe_from_fr(fr_load(p)); e1 = fe_from_fr(fr_load(p + 2)); e2 = fe_from_fr(fr_load(p + 4)); e3 = fe_from_fr(fr_load(p + 6));
            }
            const Fe l0 = fe_add(e0, fe_mul_bind<kChain>(fe_sub(e1, e0), r));
            asm volatile("" : "+v"(e3.l[8]) : "v"(l0.l[8])); // one product at a time: interleaving the two doubles the live constants
            const Fe h0 = fe_add(e2, fe_mul_bind<kChain>(fe_sub(e3, e2), r));
            if (sl.dst_top || (mode == 3 && stop)) {
                // internal F29 tables: ONE parallel carry pass, no modular reduction.  The value moves by < p + 2^231 per bind
                // (fe_mul_bind: r*(e1-e0) comes back in (-p - 2^230, 2^230)), i.e. stays within (rounds+1) p < 2^261 in magnitude for any
                // nv <= 40, which every consumer tolerates: the multipliers' bounds depend on limb sizes only (limbs 0..7 are
                // re-tightened here, the top limb stays below 2^28), and fe_to_fr reduces any |v| < 2^260 exactly.
                lo_out = fe_carry_pass(l0);
                hi_out = fe_carry_pass(h0);
                if (mode == 1) {
                    fe_store_f29(sl.dst, 2 * b, lo_out);
                    fe_store_f29(sl.dst, 2 * b + 1, hi_out);
                    *reinterpret_cast<int2 *>(sl.dst_top + 2 * b) = make_int2(lo_out.l[8], hi_out.l[8]);
                }
            } else { // tables stay canonical in the reference layout
                const Fr lc = fe_to_fr(l0), hc = fe_to_fr(h0);
                if (mode == 1) {
                    uint4 *q = sl.dst + 4 * b;
                    fr_store(q, lc);
                    fr_store(q + 2, hc);
                }
                lo_out = fe_from_fr(lc);
                hi_out = fe_from_fr(hc);
            }
        }
    }
};


} // namespace scd
