// kernels_tail.hip -- the latency-bound rounds of a whole proof with the tables RESIDENT IN LDS (k_tail_slices).
//
// k_tail_rounds (kernels.hip) walks the last rounds of a proof (<= 2048 pairs) in one launch, but every round still moves its tables
// through memory twice (bind -> grid barrier -> sums -> grid barrier -> finalize): five cross-XCD hand-overs and two trips through
// L2 / memory per round, 21-25 us for a few microseconds of arithmetic (profiles/r3j_tail_clocks.txt).  Binding is LSB-first
// (reference prover.rs:119-120 pairs entries 2b, 2b+1), so a block that owns a CONTIGUOUS range of every table owns the bound range
// too: here block g of B loads entries [g E, (g + 1) E) of every table into LDS once -- as nine 29-bit limbs, the form the products
// take -- and then needs nothing from any other block ever again:
//   every block   polls the host-mapped mailbox for the round's challenge itself (the poll is the fetch: tagged words);
//                 binds its slice in place in LDS; multiplies out every (product, node) combination over its pairs; hands its
//                 n_combos sums to block 0 as self-validating words (value | round tag << 32: no flag, no fence);
//   block 0       adds the B partials of every combination, forms the message (finalize_message), publishes it to the host.
// One hand-over per round (block -> block 0) instead of five, no table traffic at all.  When a block is down to one entry per table
// (fewer pairs than blocks), every block ships that entry to block 0 and leaves; block 0 finishes the proof alone, still out of LDS.
// At the end whoever still holds entries writes them back in the reference layout (sc_prover_state reads the final tables).
//
// Same boundary as k_tail_rounds: the message of tail round j goes to the host-mapped page with sequence seq0 + j; the challenge of
// round j >= 1 arrives in mailbox slot (sig0 + j) & 1 as eight words (limb << 32 | tag); every wait is bounded, and a block whose wait
// expires raises the give-up marker (the host voids the proof) and the device-side stop word (the other blocks leave at their next poll).
#include "finalize_device.hpp"
#include "kernel_common.hpp"

#include <algorithm>

namespace scd {

constexpr uint32_t kTsStop = 3; // word of TailArgs::sync: non-zero = leave

// an element in LDS: nine limbs in a 48-byte, 16-byte aligned slot (three ds_read_b128 / ds_write_b128 instead of nine 32-bit accesses)
constexpr int kTsEnt = 12; // dwords per entry
__device__ __forceinline__ Fe ts_lds_load(const int32_t *t) {
    const int4 a = *reinterpret_cast<const int4 *>(t), b = *reinterpret_cast<const int4 *>(t + 4), c = *reinterpret_cast<const int4 *>(t + 8);
    Fe r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x;
    return r;
}
__device__ __forceinline__ void ts_lds_store(int32_t *t, const Fe &v) {
    *reinterpret_cast<int4 *>(t) = make_int4(v.l[0], v.l[1], v.l[2], v.l[3]);
    *reinterpret_cast<int4 *>(t + 4) = make_int4(v.l[4], v.l[5], v.l[6], v.l[7]);
    t[8] = v.l[8];
}
__device__ __forceinline__ Fe fe_shfl_down(const Fe &a, const int off) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = __shfl_down(a.l[i], off, 64);
    return r;
}
// Hand-over words ("granules"): value | tag << 32, eight bytes, self-validating.  They travel two at a time: ONE 16-byte write-through
// (sc1) store per lane -- a narrower sc1 store is a fabric write of its own and costs 2.7x (8 B) to 6x (4 B) per byte -- and 16-byte sc1
// loads, both through a buffer descriptor over the hand-over area (aux = 16 = sc1); an 8-byte half is never torn.
typedef uint32_t ts_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ts_store_pair(const __amdgpu_buffer_rsrc_t rsrc, const uint32_t word_index, const uint32_t v0, const uint32_t v1, const uint32_t tag) {
    const ts_v4 v = {v0, tag, v1, tag};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)(word_index * 8u), 0, 16);
}
__device__ __forceinline__ ts_v4 ts_load_pair(const __amdgpu_buffer_rsrc_t rsrc, const uint32_t word_index) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(word_index * 8u), 0, 16);
}
#ifdef SC_TAIL_CLOCKS // tools/build_variant.sh tail_clocks -DSC_TAIL_CLOCKS: where a round goes (100 MHz wall clock, block 0; tools/tail_slices_clocks.py)
__device__ uint64_t g_ts_clk[64 * 8];
#define TS_STAMP(j, i)                                                                                                                    \
    do {                                                                                                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (j) < 64) g_ts_clk[8 * (j) + (i)] = wall_clock64();                                    \
    } while (0)
#else
#define TS_STAMP(j, i)
#endif

// kSlots: distinct tables a product may have (kMaxFusedM; kMaxWideM for lists with products of nine to twelve multiplicands, whose sums then
// carry the 2^(-5(M-1)) of the carry-free products too: finalize_message's `scaled` bit 2)
template <int kSlots>
__global__ __launch_bounds__(kTsBlock) void k_tail_slices(const TailSlicesArgs S, const ComboMeta meta, const FinMeta fin) {
    const TailArgs &A = S.base;
    extern __shared__ uint4 dyn_lds[];
    uint4 *fin_lds = dyn_lds;                                                                     // finalize_message's scratch
    int32_t *tabs = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(dyn_lds) + S.fin_bytes); // [table][entry][kTsEnt]
    __shared__ uint64_t r_sh[4];
    __shared__ uint32_t stop_sh;
    __shared__ Combo combo_sh[kMetaCombos];
    __shared__ uint32_t slot_table_sh[kMetaSlots], slot_exp_sh[kMetaSlots];
    __shared__ int prod_index_sh[kMetaCombos];
    __shared__ uint32_t part_sh[kMetaCombos * 8]; // this block's sums of the round, canonical words, on their way out
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(S.xw, 0, (int)(kTsXwWords * 8), 0x00020000); // (kernel argument: uniform)
    const int tid = threadIdx.x;
    uint32_t g = blockIdx.x, B = gridDim.x; // this block's index among the B blocks still at work (both shrink when blocks merge)
    const int U = A.n_tables;
    const uint32_t cap = S.lds_entries; // entries per table the LDS area holds
    for (int i = tid; i < kMetaCombos; i += kTsBlock) {
        combo_sh[i] = meta.combo[i];
        int k = 0;
        if (i < A.n_combos)
            while (k < A.K - 1 && fin.prod[k].partial_off != meta.combo[i].partial_off) ++k;
        prod_index_sh[i] = k;
    }
    for (int i = tid; i < kMetaSlots; i += kTsBlock) {
        slot_table_sh[i] = meta.slot_table[i];
        slot_exp_sh[i] = meta.slot_exp[i];
    }
    if (tid == 0) stop_sh = 0;
    auto prod_of = [&](int k) -> FinProd { return fin.prod[k]; };
    auto tab_at = [&](int u, uint32_t e) -> int32_t * { return tabs + ((uint32_t)u * cap + e) * (uint32_t)kTsEnt; }; // (32-bit index arithmetic: LDS)
    const uint32_t wait_spins = A.max_spins > (1u << 20) ? A.max_spins : (1u << 20); // hand-overs between blocks: bounded, but never by the patience for the host
    uint32_t *stop_word = A.sync + kTsStop;
    uint32_t *giveup = A.sig + 1;
    auto give_up = [&](uint32_t marker) { // (one lane) tell the host, tell the other blocks
        __hip_atomic_store(giveup, marker, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(stop_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // lanes per (product, node) combination: a power of two, all of a combination's lanes in one wavefront
    int L = 64;
    while (L * A.n_combos > kTsBlock) L >>= 1;
    const int my_combo = tid / L, my_q = tid % L;
    const bool combo_live = my_combo < A.n_combos;
    const uint32_t wpb = (uint32_t)A.n_combos * 8; // words of a block's sums
    __syncthreads(); // (the metadata above)
    // this lane's combination, once: its node and, per slot, the table's LDS base (in entries) and the multiplicity
    const Combo my_c = combo_sh[combo_live ? my_combo : 0];
    const int32_t my_nv = node_value((int)my_c.t);
    uint32_t my_base[kSlots], my_exp[kSlots];
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) {
        const bool in = (uint32_t)sl < my_c.n_slots;
        my_base[sl] = in ? slot_table_sh[my_c.slot_off + sl] * cap : 0u;
        my_exp[sl] = in ? slot_exp_sh[my_c.slot_off + sl] : 0u;
    }

    uint64_t n_pairs = A.first_pairs; // pairs of the round in hand, over all blocks
    uint32_t E = 0;                   // entries per table this block holds
    int binds = 0;
    for (int j = 0; j < A.n_rounds; ++j, n_pairs >>= 1) {
        const uint32_t tag = S.tag0 + (uint32_t)j;
        const bool has_bind = j > 0 || A.first_has_bind;
        TS_STAMP(j, 0); // round start
        // ---- the round's challenge: with the launch (round 0), or from the mailbox ---------------------------------------------------
        if (has_bind) {
            if (j == 0) {
                if (tid < 4) r_sh[tid] = A.r0.l[tid];
            } else if (tid < 64) {
                const uint32_t want = A.sig0 + (uint32_t)j;
                uint64_t w = 0;
                bool seen = false;
                if (S.mail_vram) {
                    // the mailbox is in device memory (the host stores into it over the BAR): every block polls it itself, locally
                    for (uint32_t spin = 0; spin < 8 * A.max_spins; ++spin) { // (a local poll is ~10x shorter than one over PCIe: same patience)
                        if (tid < 8) w = __hip_atomic_load(S.mail_vram + 8 * (want & 1u) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const bool mine = tid >= 8 || (uint32_t)w == want;
                        if (__all(mine)) { seen = true; break; }
                        if ((spin & 255u) == 255u && __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                } else if (g == 0) {
                    // block 0 asks the host (one poller: many blocks polling over PCIe queue up behind each other's reads); the others take
                    // the challenge from eight tagged words block 0 leaves in device memory -- again the poll is the fetch
                    for (uint32_t spin = 0; spin < A.max_spins; ++spin) {
                        if (tid < 8) w = __hip_atomic_load(A.mail_host + 8 * (want & 1u) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const bool mine = tid >= 8 || (uint32_t)w == want;
                        if (__all(mine)) { seen = true; break; }
                        if (__any(tid == 0 && (uint32_t)w == (want ^ 0x80000000u))) break; // the host asks the kernel to leave
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (B > 1) { // limbs 2l, 2l+1 -> lane l < 4: one 16-byte store each
                        const uint32_t limb = (uint32_t)(w >> 32);
                        const uint32_t a = __shfl(limb, 2 * (tid & 3), 64), b2 = __shfl(limb, 2 * (tid & 3) + 1, 64);
                        if (seen && tid < 4) ts_store_pair(xrs, (uint32_t)(kTsXwWords - 8) + 2 * (uint32_t)tid, a, b2, tag);
                    }
                } else {
                    const uint64_t *bc = S.xw + kTsXwWords - 8;
                    for (uint32_t spin = 0; spin < wait_spins; ++spin) {
                        if (tid < 8) w = __hip_atomic_load(bc + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const bool mine = tid >= 8 || (uint32_t)(w >> 32) == tag;
                        if (__all(mine)) { seen = true; break; }
                        if ((spin & 31u) == 31u && __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    w <<= 32; // (the limb in the high half, as the host's words carry it)
                }
                if (!seen && tid == 0) {
                    give_up(want);
                    stop_sh = 1;
                }
                const uint32_t lo32 = (uint32_t)(w >> 32);
                const uint32_t hi32 = __shfl_down(lo32, 1, 64);
                if (tid < 8 && (tid & 1) == 0) r_sh[tid >> 1] = (uint64_t)lo32 | ((uint64_t)hi32 << 32);
            }
            __syncthreads();
            // no challenge (the interactive protocol's verifier took longer than the kernel's patience, or the host asked it to leave):
            // every block is between two rounds, its slice as round j - 1 left it -- written back below, so the handle carries on from there
            if (stop_sh) break;
        }
        TS_STAMP(j, 1); // challenge in hand
        FeU r32;
        if (has_bind) {
            FrHost rh;
#pragma unroll
            for (int i = 0; i < 4; ++i) rh.l[i] = r_sh[i];
            r32 = feu_shl5(fru_from_host(rh).v); // the carry-free bind's multiplier: r * 2^5 as 29-bit limbs
        }
        // ---- the slice: loaded from the caller's / the big rounds' tables in round 0 (bound on the way in), bound in place afterwards ----
        if (j == 0) {
            E = (uint32_t)((2 * n_pairs) / B); // entries per table and block of THIS round's tables
            const uint32_t total = E * (uint32_t)U;
            const int shE = 31 - __builtin_clz(E); // (E is a power of two: no integer divisions on a latency-bound path)
            for (uint32_t i = tid; i < total; i += kTsBlock) {
                const uint32_t u = i >> shE, e = i & (E - 1);
                const uint64_t ge = (uint64_t)g * E + e; // the entry's index in this round's table
                const uint4 *src = A.t.cur0[u];
                const int32_t *stop = A.t.cur0_top[u];
                Fe v;
                if (has_bind) { // entries 2 ge, 2 ge + 1 of the previous table
                    Fe lo, hi;
                    if (stop) {
                        const int2 t = *reinterpret_cast<const int2 *>(stop + 2 * ge);
                        lo = fe_load_f29(src, 2 * ge, t.x);
                        hi = fe_load_f29(src, 2 * ge + 1, t.y);
                    } else {
                        lo = fe_from_fr(fr_load(src + 4 * ge));
                        hi = fe_from_fr(fr_load(src + 4 * ge + 2));
                    }
                    v = fe_carry_pass(fe_add(lo, fe_mul_u<true>(fe_sub(hi, lo), r32)));
                } else {
                    v = stop ? fe_load_f29(src, ge, stop[ge]) : fe_from_fr(fr_load(src + 2 * ge));
                }
                ts_lds_store(tab_at((int)u, e), v);
            }
            if (has_bind) binds += 1;
            __syncthreads();
        } else {
            // in place: entry e <- entries 2e, 2e + 1.  A pass reads everything it needs before it writes (one barrier); later passes
            // read higher entries than any earlier pass wrote (2 e'' > e for e'' > e).
            const uint32_t half = E / 2, total = half * (uint32_t)U;
            const int shH = 31 - __builtin_clz(half);
            for (uint32_t i0 = 0; i0 < total; i0 += kTsBlock) {
                const uint32_t i = i0 + tid;
                const bool live = i < total;
                const uint32_t u = live ? i >> shH : 0, e = live ? i & (half - 1) : 0;
                Fe v = fe_zero();
                if (live) {
                    const Fe lo = ts_lds_load(tab_at((int)u, 2 * e)), hi = ts_lds_load(tab_at((int)u, 2 * e + 1));
                    v = fe_carry_pass(fe_add(lo, fe_mul_u<true>(fe_sub(hi, lo), r32)));
                }
                __syncthreads();
                if (live) ts_lds_store(tab_at((int)u, e), v);
            }
            E = half;
            binds += 1;
            __syncthreads();
        }
        // ---- fewer pairs than blocks: groups of R adjacent blocks ship their last entry of every table to the group's first block
        // and leave; the survivors hold R entries each and carry on (R = 16, or all that are left) ----------------------------------------
        if (B > 1 && E == 1) {
            const uint32_t R = B < 16u ? B : 16u;
            const int shR = 31 - __builtin_clz(R);
            // (an entry = nine limbs + one filler word = five 16-byte pairs; layout [table][block][10 words])
            for (int i = tid; i < U * 5; i += kTsBlock) {
                const int u = i / 5, pr = i % 5;
                const int32_t *e0 = tab_at(u, 0);
                ts_store_pair(xrs, (uint32_t)kTsAccWords + ((uint32_t)u * B + g) * 10 + 2 * (uint32_t)pr, (uint32_t)e0[2 * pr], pr < 4 ? (uint32_t)e0[2 * pr + 1] : 0u, tag);
            }
            if ((g & (R - 1)) != 0) return;
            __syncthreads(); // (the leader's own entry 0 of every table is about to be overwritten by the same value: harmless, but ordered)
            // (requests in batches of kXferBatch pairs per lane, all in flight before the first is looked at: one trip to memory per batch)
            bool ok = true;
            constexpr int kXferBatch = 12;
            const uint32_t per_table = R * 5, total_p = (uint32_t)U * per_table; // pairs: [table][member][5]
            for (uint32_t c0 = 0; c0 < total_p && ok; c0 += kXferBatch * kTsBlock) {
                ts_v4 w[kXferBatch];
                bool got = false;
                for (uint32_t spins = 0; spins <= wait_spins && !got; ++spins) {
                    got = true;
#pragma unroll
                    for (int k = 0; k < kXferBatch; ++k) {
                        if (c0 + (uint32_t)k * kTsBlock < total_p) {
                            const uint32_t i = min(c0 + (uint32_t)k * kTsBlock + (uint32_t)tid, total_p - 1);
                            const uint32_t ent = i / 5, pr = i - 5 * ent; // ent = table * R + member
                            w[k] = ts_load_pair(xrs, (uint32_t)kTsAccWords + ((ent >> shR) * B + g + (ent & (R - 1))) * 10 + 2 * pr);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < kXferBatch; ++k)
                        if (c0 + (uint32_t)k * kTsBlock < total_p) got = got && w[k].y == tag && w[k].w == tag;
                    if (!got) {
                        if ((spins & 63u) == 63u && __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                ok = got;
#pragma unroll
                for (int k = 0; k < kXferBatch; ++k) {
                    if (c0 + (uint32_t)k * kTsBlock < total_p) {
                        const uint32_t i = min(c0 + (uint32_t)k * kTsBlock + (uint32_t)tid, total_p - 1);
                        const uint32_t ent = i / 5, pr = i - 5 * ent;
                        int32_t *dst = tab_at((int)(ent >> shR), ent & (R - 1));
                        dst[2 * pr] = (int32_t)w[k].x;
                        if (pr < 4) dst[2 * pr + 1] = (int32_t)w[k].z;
                    }
                }
            }
            if (!ok) stop_sh = 1;
            __syncthreads();
            if (stop_sh) {
                if (tid == 0) give_up(A.sig0 + (uint32_t)j + 1u);
                return;
            }
            E = R;
            g >>= shR;
            B >>= shR;
        }
        TS_STAMP(j, 2); // slice bound (and, where blocks merged, collected)
        // ---- sums: lane (combination, q) multiplies out the combination's pairs q, q + L, ... of this block ---------------------------
        const uint32_t pairs_here = E / 2;
        Fe acc = fe_zero();
        if (combo_live) {
            const int32_t nv = my_nv;
            uint32_t iter = 0;
            for (uint32_t pr = (uint32_t)my_q; pr < pairs_here; pr += (uint32_t)L, ++iter) {
                Fe prod = fe_zero();
                bool first = true;
#pragma unroll
                for (int sl = 0; sl < kSlots; ++sl) {
                    if (my_exp[sl] == 0) break; // (slots are dense: the first empty one ends the list)
                    const int32_t *lo_p = tabs + (my_base[sl] + 2 * pr) * (uint32_t)kTsEnt;
                    Fe val;
                    if (nv == 0) val = ts_lds_load(lo_p);
                    else if (nv == 1) val = ts_lds_load(lo_p + kTsEnt);
                    else val = fe_line(ts_lds_load(lo_p), ts_lds_load(lo_p + kTsEnt), nv);
                    uint32_t k = 0;
                    if (first) { prod = val; k = 1; first = false; }
                    for (const uint32_t e = my_exp[sl]; k < e; ++k) prod = fe_mul<true>(val, prod);
                }
                acc = fe_carry_pass(fe_add(acc, prod));
                if ((iter & 31u) == 31u) acc = fe_from_fr(fe_to_fr(acc)); // (keeps the top limb far from 2^31; never reached here)
            }
        }
        for (int off = L >> 1; off >= 1; off >>= 1) acc = fe_carry_pass(fe_add(acc, fe_shfl_down(acc, off)));
        TS_STAMP(j, 3); // this block's sums
        if (B == 1) {
            if (combo_live && my_q == 0) fr_store(fin_lds + 2 * (prod_index_sh[my_combo] * A.D + (int)combo_sh[my_combo].t), fe_to_fr(acc));
        } else {
            // ---- this block's sums -> block 0.  The canonical 32-bit words are ADDED, by the memory system, into 64-bit accumulators:
            // (1 << 44 | word) per block, so that a word whose top bits read the number of its contributors IS complete -- block 0's poll
            // of the accumulators is the fetch of the sums (a few hundred words whatever the number of blocks).  Eight groups of blocks
            // (g & 7) keep a sum below 32 p (fe_to_fr's range); a ring of four accumulator sets, the one of round j + 2 zeroed by block 0.
            if (combo_live && my_q == 0) {
                const Fr sv = fe_to_fr(acc);
#pragma unroll
                for (int i = 0; i < 8; ++i) part_sh[my_combo * 8 + i] = sv.v[i];
            }
            __syncthreads();
            const uint32_t groups = B < 8u ? B : 8u, per_group = B / groups;
            uint64_t *ring = S.xw + (size_t)(j & 3) * (8 * kMetaCombos * 8);
            if ((uint32_t)tid < wpb)
                (void)__hip_atomic_fetch_add(ring + (g & (groups - 1)) * wpb + (uint32_t)tid, (1ULL << 44) | (uint64_t)part_sh[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g == 0) {
                constexpr int kGatherMax = (8 * kMetaCombos * 4 + kTsBlock - 1) / kTsBlock; // pairs of words per lane
                const uint32_t total_p = groups * wpb / 2, ring_w0 = (uint32_t)(j & 3) * (8 * kMetaCombos * 8);
                uint64_t *stage = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(dyn_lds) + S.stage_off); // [group][combination * 8 + word]
                ts_v4 w[kGatherMax];
                bool got = false;
                for (uint32_t spins = 0; spins <= wait_spins && !got; ++spins) {
                    bool mine = true;
                    // (no per-lane predication: a lane past the end re-reads the last pair -- clamped address -- and only wave-uniform
                    // branches skip the unused passes)
#pragma unroll
                    for (int k = 0; k < kGatherMax; ++k) {
                        if ((uint32_t)k * kTsBlock < total_p) w[k] = ts_load_pair(xrs, ring_w0 + 2 * min((uint32_t)k * kTsBlock + (uint32_t)tid, total_p - 1));
                    }
#pragma unroll
                    for (int k = 0; k < kGatherMax; ++k)
                        if ((uint32_t)k * kTsBlock < total_p) mine = mine && (w[k].y >> 12) == per_group && (w[k].w >> 12) == per_group;
                    got = __all(mine) != 0;
                    if (!got) {
                        if ((spins & 63u) == 63u && __hip_atomic_load(stop_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (!got) stop_sh = 1; // (benign race: every writer stores 1)
                TS_STAMP(j, 6); // every block's words are in (this wave's share)
#pragma unroll
                for (int k = 0; k < kGatherMax; ++k) {
                    if ((uint32_t)k * kTsBlock < total_p) { // (lanes past the end store the last pair again)
                        const uint32_t pi = min((uint32_t)k * kTsBlock + (uint32_t)tid, total_p - 1);
                        stage[2 * pi] = (uint64_t)w[k].x | ((uint64_t)(w[k].y & 0xfffu) << 32);
                        stage[2 * pi + 1] = (uint64_t)w[k].z | ((uint64_t)(w[k].w & 0xfffu) << 32);
                    }
                }
                __syncthreads();
                Fr *hsum = reinterpret_cast<Fr *>(stage + groups * wpb); // [group][combination]
                if ((uint32_t)tid < groups * (uint32_t)A.n_combos) {
                    // sum_q S_q 2^(32 q), S_q < 2^37: carry through eight 32-bit words; what is left over (< 2^6) sits at 2^256 = 2^24 in limb 8
                    Fr x;
                    uint64_t carry = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const uint64_t t = stage[(uint32_t)tid * 8 + q] + carry; // (group-major, combination, word: the same order as the lanes)
                        x.v[q] = (uint32_t)t;
                        carry = t >> 32;
                    }
                    Fe v = fe_from_fr(x);
                    v.l[8] += (int32_t)(carry << 24);
                    hsum[tid] = fe_to_fr(v);
                }
                TS_STAMP(j, 7); // folded per group
                __syncthreads();
                if (tid < A.n_combos) {
                    Fr r = hsum[tid];
                    for (uint32_t q = 1; q < groups; ++q) r = fr_add(r, hsum[q * (uint32_t)A.n_combos + (uint32_t)tid]);
                    fr_store(fin_lds + 2 * (prod_index_sh[tid] * A.D + (int)combo_sh[tid].t), r);
                }
            }
        }
        if (g == 0) {
            __syncthreads();
            if (stop_sh) {
                if (tid == 0) give_up(A.sig0 + (uint32_t)j + 1u);
                return;
            }
            TS_STAMP(j, 4); // node sums complete (every block's in)
            finalize_message<kTsBlock>(prod_of, A.Wm, A.K, A.D, fin_lds, (uint4 *)nullptr, (uint64_t *)nullptr, A.h_out, A.h_flag, A.seq0 + (uint32_t)j, kSlots > kMaxFusedM ? 5 : 1, (const Fr *)nullptr);
            TS_STAMP(j, 5); // message published
            if (B > 1) { // the accumulators of round j + 2 (last used in round j - 2: complete and read long ago)
                const uint32_t z0 = (uint32_t)((j + 2) & 3) * (8 * kMetaCombos * 8);
                for (uint32_t i = tid; i < 8 * kMetaCombos * 4; i += kTsBlock) ts_store_pair(xrs, z0 + 2 * i, 0u, 0u, 0u);
            }
            __syncthreads(); // (fin_lds is written again by the next round's sums)
        }
    }
    // ---- the tables as the rounds left them, in the reference layout, where the handle expects them after `binds` binds ------------------
    if (binds > 0) {
        const uint32_t total = E * (uint32_t)U;
        const int shE = 31 - __builtin_clz(E);
        for (uint32_t i = tid; i < total; i += kTsBlock) {
            const uint32_t u = i >> shE, e = i & (E - 1);
            uint4 *dst = (binds & 1) ? A.t.b0[u] : A.t.b1[u];
            fr_store(dst + 2 * ((uint64_t)g * E + e), fe_to_fr(ts_lds_load(tab_at((int)u, e))));
        }
    }
}

// LDS a launch needs: finalize's scratch + the table area (entries per table: a block's slice, and the sixteen a merge leaves its survivors)
// + block 0's staging of the accumulators
static size_t ts_lds_entries(uint64_t first_pairs, int B) { return (size_t)std::max<uint64_t>(2 * first_pairs / (uint64_t)B, B > 1 ? 16 : 0); }
static size_t ts_fin_bytes(int K, int D) { return ((size_t)K * D * (D + 2) * 32 + 15) & ~(size_t)15; }
static size_t ts_stage_bytes(int n_combos, int B) { return B > 1 ? 8 * (size_t)n_combos * 64 + 8 * (size_t)n_combos * 32 + 16 : 0; }
constexpr size_t kTsLdsMax = 144 * 1024; // of the CU's 160 KB (one block per CU; its static LDS is ~3 KB)

int tail_slices_blocks(uint64_t first_pairs, int n_tables, int K, int D, int n_combos, int max_multiplicands, int max_blocks) {
    if (max_blocks < 1) return 0;
    if (first_pairs == 0 || (first_pairs & (first_pairs - 1)) != 0 || first_pairs > kTsMaxPairs) return 0;
    if (first_pairs > kSmallRoundPairs) {
        // Above the big / small round boundary the fused tree kernels stream the tables at memory speed; a resident block, one wavefront
        // per SIMD, beats them only while its lanes have few dependent products in a row: passes over its pairs x multiplicands <= 12
        // (config 2's and the GKR phases' shapes up to 2^16 pairs, two products of three up to 2^15; config 3's shape not at all).
        int L = 64;
        while (L * n_combos > kTsBlock) L >>= 1;
        const uint64_t pairs_b = first_pairs / kTsMaxBlocks, passes = (pairs_b + L - 1) / L;
        if (passes * (uint64_t)max_multiplicands > 12) return 0;
    }
    // as many blocks as leave every block two hand-over rounds (B <= first_pairs / 2), one block for short tails
    int B = kTsMaxBlocks;
    while (B > 1 && (uint64_t)B > first_pairs / 2) B >>= 1;
    if (first_pairs <= 32) B = 1;
    while (B > max_blocks) B >>= 1; // (every block must be resident: they hand over to each other)
    const size_t bytes = ts_fin_bytes(K, D) + ts_lds_entries(first_pairs, B) * (size_t)n_tables * (kTsEnt * 4) + ts_stage_bytes(n_combos, B);
    return bytes <= kTsLdsMax ? B : 0; // (more blocks than kTsMaxBlocks would be needed: the caller runs this round as launches and asks again)
}

hipError_t ensure_dynamic_lds(const void *kernel, int bytes, bool (&done)[64]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    bool &flag = done[(unsigned)dev & 63u]; // (a benign race: two threads of one device may both set the attribute)
    if (__atomic_load_n(&flag, __ATOMIC_ACQUIRE)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_store_n(&flag, true, __ATOMIC_RELEASE);
    return e;
}
template <int kSlots>
static hipError_t tail_slices_attr() { // (more dynamic LDS than the default 64 KB limit of a launch)
    static bool done[64] = {};
    return ensure_dynamic_lds(reinterpret_cast<const void *>(k_tail_slices<kSlots>), (int)kTsLdsMax, done);
}
template <int kSlots>
static int tail_slices_max_blocks_t(int device) {
    static int cached[64] = {}; // 0: not asked yet; -1: unknown
    int &c = cached[(unsigned)device & 63u];
    if (c != 0) return c > 0 ? c : 0;
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (tail_slices_attr<kSlots>() != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tail_slices<kSlots>, kTsBlock, kTsLdsMax) != hipSuccess || per_cu < 1 ||
        hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount < 1) {
        (void)hipGetLastError();
        c = -1;
        return 0;
    }
    c = per_cu * prop.multiProcessorCount;
    return c;
}
int tail_slices_max_blocks(int device, int max_multiplicands) {
    return max_multiplicands > kMaxFusedM ? tail_slices_max_blocks_t<kMaxWideM>(device) : tail_slices_max_blocks_t<kMaxFusedM>(device);
}
template <int kSlots>
static hipError_t launch_tail_slices_t(const TailSlicesArgs &args, const ComboMeta &meta, const FinMeta &fin, size_t lds, hipStream_t stream) {
    hipError_t e = tail_slices_attr<kSlots>();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_tail_slices<kSlots>, dim3(args.B), dim3(kTsBlock), lds, stream, args, meta, fin);
    return hipGetLastError();
}

hipError_t launch_tail_slices(TailSlicesArgs args, const ComboMeta &meta, const FinMeta &fin, int max_multiplicands, hipStream_t stream) {
    const int B = args.B;
    if (B < 1 || B > kTsMaxBlocks || (B & (B - 1)) != 0) return hipErrorInvalidValue;
    args.fin_bytes = (uint32_t)ts_fin_bytes(args.base.K, args.base.D);
    args.lds_entries = (uint32_t)ts_lds_entries(args.base.first_pairs, B);
    args.stage_off = (uint32_t)((args.fin_bytes + (size_t)args.lds_entries * args.base.n_tables * (kTsEnt * 4) + 15) & ~(size_t)15);
    const size_t lds = args.stage_off + ts_stage_bytes(args.base.n_combos, B);
    if (lds > kTsLdsMax) return hipErrorInvalidValue;
    if (max_multiplicands > kMaxWideM) return hipErrorInvalidValue;
    return max_multiplicands > kMaxFusedM ? launch_tail_slices_t<kMaxWideM>(args, meta, fin, lds, stream) : launch_tail_slices_t<kMaxFusedM>(args, meta, fin, lds, stream);
}

} // namespace scd

#ifdef SC_TAIL_CLOCKS
extern "C" __attribute__((visibility("default"))) int sc_debug_tail_slices_clocks(uint64_t *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scd::g_ts_clk), sizeof(uint64_t) * 64 * 8);
}
#endif
