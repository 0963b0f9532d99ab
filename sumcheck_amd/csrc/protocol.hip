// protocol.hip -- the per-round launch plan and the host-side protocol drivers that the reference runs around prove_round
// (reference src/ml_sumcheck/mod.rs:50-70): sc_prove_round (launches, the resident kernel of the interactive protocol), the
// Fiat-Shamir loop with pipelined late rounds and the persistent tail kernel, sc_ml_prove*, and the sharded proof of one rank.
#include "prover_internal.hpp"

// Launch one round's kernels on p->stream.  On return the round polynomial is in p->d_out (and in
// d_wide if non-null); nothing has been synchronised.
uint64_t small_pairs_limit() { // the big/small round boundary
#ifdef SC_EXPERIMENTS // SC_SMALL_LOG2
    static const uint64_t v = [] {
        const char *e = std::getenv("SC_SMALL_LOG2");
        return e ? (1ULL << std::atoi(e)) : scd::kSmallRoundPairs;
    }();
    return v;
#else
    return scd::kSmallRoundPairs;
#endif
}

// One-time probe per process: does a kernel launch return before the kernel has finished?  A wait kernel with a short bound
// (a few milliseconds) is enqueued on a word nobody sets; an asynchronous runtime returns from the launch call at once, a
// serialising one (a profiler collecting counters, *_LAUNCH_BLOCKING) only when the bound has expired -- and then pipelined
// rounds, whose wait kernels must be enqueued BEFORE the host produces the challenge, are not possible.
bool launches_are_async(sc_prover *p) {
    static const bool ok = [p] {
        uint32_t *h = nullptr, *d = nullptr;
        FrHost *dm = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&h), 256, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return false;
        bool good = hipHostGetDevicePointer(reinterpret_cast<void **>(&d), h, 0) == hipSuccess &&
                    hipMalloc(reinterpret_cast<void **>(&dm), sizeof(FrHost)) == hipSuccess;
        if (good) {
            std::memset(h, 0, 256);
            // (a first launch of the process also loads the code object: milliseconds that say nothing about the launch mode)
            good = scd::launch_wait_challenge(d, 0xffffffffu, reinterpret_cast<const FrHost *>(d + 16), dm, p->stream, 1) == hipSuccess &&
                   hipStreamSynchronize(p->stream) == hipSuccess;
            __atomic_store_n(h + 1, 0u, __ATOMIC_RELEASE);
            const auto t0 = std::chrono::steady_clock::now();
            good = good && scd::launch_wait_challenge(d, 0xffffffffu, reinterpret_cast<const FrHost *>(d + 16), dm, p->stream, 1u << 11) == hipSuccess;
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            (void)hipStreamSynchronize(p->stream);
            good = good && ms < 1.0; // 2^11 polls take a few milliseconds; an asynchronous launch call a few microseconds
        }
        if (dm) (void)hipFree(dm);
        (void)hipHostFree(h);
        (void)hipGetLastError();
        return good;
    }();
    return ok;
}

// The host-mapped mailbox (two challenge slots + the signal word the device polls) of the pipelined rounds and of the persistent
// tail kernel.  First use sets it up; any failure -- or a runtime that serialises launches, or sc_set_policy("pipeline", 0) -- switches both off
// for this handle.
bool ensure_mailbox(sc_prover *p) {
    if (!p->pipeline_ok) return false;
    if (p->sig) return true;
    bool env_off = scd::policy(scd::kPolPipeline) == 0; // read per handle, at its first late round
    // a runtime that makes every launch wait for its kernel would block on the waiting kernel until its bound expires
    for (const char *name : {"AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"}) {
        const char *v = std::getenv(name);
        if (v && std::atoi(v) != 0) env_off = true;
    }
    bool ok = !env_off && hipSetDevice(p->device) == hipSuccess && launches_are_async(p);
    // layout: [0, 64) two challenge slots (k_wait_challenge) | [64, 128) signal word + give-up marker | [128, 256) two slots of eight
    // tagged 64-bit words (k_tail_rounds)
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&p->h_mail), 256, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok) std::memset(p->h_mail, 0, 256);
    ok = ok && hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_mail_dev), p->h_mail, 0) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void **>(&p->d_mail), 2 * sizeof(FrHost)) == hipSuccess;
    if (ok) {
        p->sig = reinterpret_cast<uint32_t *>(p->h_mail + 2);
        p->sig_dev = reinterpret_cast<uint32_t *>(p->h_mail_dev + 2);
        __atomic_store_n(p->sig, 0u, __ATOMIC_RELEASE);
        __atomic_store_n(p->sig + 1, 0u, __ATOMIC_RELEASE); // give-up marker of the waiting kernel
        p->sig_seq = 0;
        return true;
    }
    (void)hipGetLastError();
    if (p->h_mail) (void)hipHostFree(p->h_mail);
    if (p->d_mail) (void)hipFree(p->d_mail);
    p->sig = nullptr;
    p->h_mail = nullptr;
    p->d_mail = nullptr;
    p->pipeline_ok = false;
    return false;
}
// Pipelined late rounds.  can_defer_next: the NEXT round is a latency-bound one and the mailbox machinery is available.
// SC_FIN_MB=0 (experiments build): the single-block finalize
// node 1 from the claim identity in the big binding rounds (kernels.h: ClaimArgs); -DSC_NO_SKIP1: A/B build without it
bool skip1_enabled() {
#ifdef SC_NO_SKIP1
    return false;
#elif defined(SC_EXPERIMENTS)
    static const bool on = !(std::getenv("SC_SKIP1") && std::atoi(std::getenv("SC_SKIP1")) == 0);
    return on;
#else
    return true;
#endif
}
// products of five to eight multiplicands as a product tree with node extension (kernels_wide.hip); sc_set_policy("wide_tree", 0): node by node (k_prod_round_fe)
bool wide_tree_enabled() {
    return scd::policy(scd::kPolWideTree) != 0;
}
bool fin_mb_enabled() {
#ifdef SC_EXPERIMENTS
    static const bool on = !(std::getenv("SC_FIN_MB") && std::atoi(std::getenv("SC_FIN_MB")) == 0);
    return on;
#else
    return true;
#endif
}
bool can_defer_next(sc_prover *p) {
    if (!p->pipeline_ok || p->exhausted || p->round == 0 || p->round >= p->nv) return false;
    if (p->streamed && p->round < 2) return false; // round 2 of a streamed handle walks the host tables chunk by chunk
    const uint64_t n_pairs_next = 1ULL << (p->nv - (p->round + 1));
    if (!(n_pairs_next <= small_pairs_limit() && p->U <= (uint32_t)scd::kMaxSmallTables && p->K > 0)) return false;
    return ensure_mailbox(p);
}
// the challenge of the round enqueued with deferred = true: mailbox first, then the signal the stream is waiting on
void provide_challenge(sc_prover *p, const sch::Fr &r) {
    p->randomness.push_back(r);
    FrHost *slot = p->h_mail + (p->sig_seq & 1u);
    std::memcpy(slot, &r, sizeof(FrHost));
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __atomic_store_n(p->sig, p->sig_seq, __ATOMIC_RELEASE);
    p->deferred_pending = false;
}
// the give-up marker of k_wait_challenge (non-zero once any wait of this handle has expired; cleared by sc_prover_reset)
bool wait_gave_up(sc_prover *p) { return p->sig && __atomic_load_n(p->sig + 1, __ATOMIC_ACQUIRE) != 0; }
// error path: let a stream that is blocked on the wait drain (the round then runs on a stale challenge; its result is discarded)
void abandon_deferred(sc_prover *p) {
    DeviceGate gate_(p->device);
    if (p->deferred_pending) {
        __atomic_store_n(p->sig, p->sig_seq, __ATOMIC_RELEASE);
        p->deferred_pending = false;
        (void)hipStreamSynchronize(p->stream);
        p->exhausted = true; // tables are no longer meaningful: the handle must be reset
    }
}

// rows (r * 2^(29 i + 58)) mod p as plain 29-bit limbs: the challenge as the tree kernels' bind takes it (fe_device.hpp, fe_mul_bind)
void make_bind_const(const sch::Fr &r, scd::BindConst &rc) {
    static const std::array<sch::Fr, 9> pow2 = [] { // Montgomery form of 2^(29 i + 58)
        std::array<sch::Fr, 9> t;
        sch::Fr c = sch::kOne;
        for (int d = 0; d < 58; ++d) c = sch::add(c, c);
        for (int i = 0; i < 9; ++i) {
            t[i] = c;
            for (int d = 0; d < 29; ++d) c = sch::add(c, c);
        }
        return t;
    }();
    for (int i = 0; i < 9; ++i) {
        const sch::Fr x = sch::to_canonical(sch::mul(r, pow2[i]));
        for (int k = 0; k < 9; ++k) {
            const int bit = 29 * k, w = bit >> 6, sh = bit & 63;
            uint64_t v = x.l[w] >> sh;
            if (sh > 35 && w < 3) v |= x.l[w + 1] << (64 - sh);
            rc.R[i][k] = (int32_t)(v & 0x1fffffffULL);
        }
    }
}

// Rounds 1 and 2 of a handle whose tables stay in host memory (SC_TABLES_STREAM).  The round is the sum of its chunks: chunk c = entries
// [c 2^L, (c+1) 2^L) of every table goes host -> staging slot c & 1 on the copy stream while the previous chunk computes; the merged
// big-round kernel runs on the slot (round 1: sums only; round 2: bind + sums, the bound half-chunk written to its place in the
// resident table), k_finalize turns the chunk's partials into a message and k_msg_accumulate adds it to the round's.  After round 2 the
// bound tables (half the input) are resident and the ordinary path takes over.
int launch_round_streamed(sc_prover *p, const uint64_t *r_or_null, uint64_t *d_wide, bool publish_to_host) {
    scd::plan_hit(scd::kPlanBigStreamed);
    if (p->exhausted) return sc_internal_fail(SC_ERR_NOT_ACTIVE, "Prover is not active");
    if (r_or_null && p->round == 0) return sc_internal_fail(SC_ERR_FIRST_ROUND_HAS_MSG, "first round should be prover first.");
    if (!r_or_null && p->round > 0) return sc_internal_fail(SC_ERR_MISSING_MSG, "verifier message is empty");
    sch::Fr r = sch::zero();
    if (r_or_null) {
        std::memcpy(&r, r_or_null, 32);
        if (sch::geq_p(r)) return sc_internal_fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
    }
    HIP_TRY(hipSetDevice(p->device));
    int rc_t = collect_timing(p);
    if (rc_t) return rc_t;
    const bool bind = r_or_null != nullptr;
    if (bind) p->randomness.push_back(r);
    p->round += 1;
    scd::BindConst rc;
    std::memset(&rc, 0, sizeof(rc));
    if (bind) make_bind_const(r, rc);
    const uint64_t C = 1ULL << p->chunk_log2, n = 1ULL << p->nv, n_chunks = n / C;
    const uint64_t pairs_per_chunk = bind ? C / 4 : C / 2; // round 2 reads four entries per pair of the bound table
    const bool merged = p->merge_rounds && !p->any_generic; // one launch per chunk (k_round_tree*); otherwise one launch per product
    const int grid = merged ? std::min(scd::grid_for_pairs(pairs_per_chunk), scd::kRoundTreeGrid) : scd::grid_for_pairs(pairs_per_chunk);
    sch::Fr r32v = sch::zero(); // (no product kernel of the per-product path binds: the chunk is bound by k_fix first)
    const FrHost r32 = to_dev(r32v);
    p->seq += 1;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const int q = (int)(c & 1);
        if (c >= 2) HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->ev_consumed[q], 0)); // the slot's previous chunk has been read
        for (uint32_t u = 0; u < p->U; ++u)
            HIP_TRY(hipMemcpyAsync(static_cast<char *>(p->ring[q]) + (((size_t)u << p->chunk_log2) * 32), p->host_tabs[u] + 4 * c * C, C * 32, hipMemcpyHostToDevice,
                                   p->copy_stream));
        HIP_TRY(hipEventRecord(p->ev_copied[q], p->copy_stream));
        HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_copied[q], 0));
        auto ring_tab = [&](uint32_t u) { return reinterpret_cast<const uint4 *>(static_cast<char *>(p->ring[q]) + (((size_t)u << p->chunk_log2) * 32)); };
        if (merged) {
            scd::RoundArgs ra;
            std::memset(&ra, 0, sizeof(ra));
            ra.n_prod = (int)p->K;
            std::vector<uint8_t> bound(p->U, 0);
            for (uint32_t k = 0; k < p->K; ++k) {
                const Product &pr = p->prods[k];
                scd::TreeProd &tp = ra.prod[k];
                tp.M = pr.M;
                tp.partial_off = pr.partial_off;
                int f = 0;
                for (size_t s = 0; s < pr.tables.size(); ++s) {
                    const uint32_t u = pr.tables[s];
                    Table &t = p->tabs[u];
                    for (uint32_t rep = 0; rep < pr.exps[s]; ++rep, ++f) {
                        scd::Slot &sl = tp.slot[f];
                        sl.exp = 1;
                        sl.src = ring_tab(u);
                        sl.src_top = nullptr;
                        if (!bind) {
                            sl.mode = 0;
                        } else if (!bound[u]) { // this chunk's half of the bound table, in place (F29 blocks of 128 entries stay aligned: C / 2 >= 512)
                            sl.mode = 1;
                            sl.dst = t.buf[0] + 2 * (c * (C / 2));
                            sl.dst_top = p->use_f29 ? t.buf_top[0] + c * (C / 2) : nullptr;
                            bound[u] = 1;
                        } else {
                            sl.mode = 3;
                            sl.dst_top = p->use_f29 ? t.buf_top[0] : nullptr;
                        }
                    }
                }
            }
            HIP_TRY(scd::launch_round_tree(ra, rc, pairs_per_chunk, p->d_partials, grid, p->stream, true));
            if (bind) { // tables no product refers to still follow the state machine
                for (uint32_t u = 0; u < p->U; ++u)
                    if (!bound[u]) HIP_TRY(scd::launch_fix(ring_tab(u), p->tabs[u].buf[0] + 2 * (c * (C / 2)), to_dev(r), C / 2, p->stream));
            }
        } else {
            // Any other shape (more than 12 products, more than four multiplicands): the chunk is bound table by table (k_fix, into its
            // place in the resident table, canonical reference layout) and every product then sums over what it needs -- the staged chunk
            // in round 1, the freshly bound half-chunk in round 2 -- with the kernel launch_round would give it.
            std::vector<const uint4 *> src(p->U);
            for (uint32_t u = 0; u < p->U; ++u) {
                if (bind) {
                    uint4 *dst = p->tabs[u].buf[0] + 2 * (c * (C / 2));
                    HIP_TRY(scd::launch_fix(ring_tab(u), dst, to_dev(r), C / 2, p->stream));
                    src[u] = dst;
                } else {
                    src[u] = ring_tab(u);
                }
            }
            bool ptrs_uploaded = false;
            for (uint32_t k = 0; k < p->K; ++k) {
                const Product &pr = p->prods[k];
                FrHost *partials = p->d_partials + pr.partial_off;
                ProdArgs a;
                std::memset(&a, 0, sizeof(a));
                if (pr.fused && p->kernel_variant == 3 && pr.M <= 4) { // product tree: one slot per FACTOR
                    a.n_slots = (int)pr.M;
                    int f = 0;
                    for (size_t s2 = 0; s2 < pr.tables.size(); ++s2)
                        for (uint32_t rep = 0; rep < pr.exps[s2]; ++rep, ++f) {
                            a.slot[f].exp = 1;
                            a.slot[f].mode = 0;
                            a.slot[f].src = src[pr.tables[s2]];
                        }
                    HIP_TRY(scd::launch_prod_tree((int)pr.M, a, rc, pairs_per_chunk, partials, grid, p->stream));
                } else if (pr.fused) { // node by node, carry-free arithmetic: one slot per distinct table
                    a.n_slots = (int)pr.tables.size();
                    for (size_t s2 = 0; s2 < pr.tables.size(); ++s2) {
                        a.slot[s2].exp = pr.exps[s2];
                        a.slot[s2].mode = 0;
                        a.slot[s2].src = src[pr.tables[s2]];
                    }
                    HIP_TRY(scd::launch_prod_round_fe((int)pr.M, a, r32, pairs_per_chunk, partials, grid, p->stream));
                } else { // any number of multiplicands: table pointers through device memory, one set per staging slot
                    if (!ptrs_uploaded) {
                        const uint4 **h = p->h_cur_tables + (size_t)q * p->U;
                        if (c >= 2) HIP_TRY(hipEventSynchronize(p->ev_consumed[q])); // the pinned set's previous upload (chunk c - 2) has been read
                        for (uint32_t u = 0; u < p->U; ++u) h[u] = src[u];
                        HIP_TRY(hipMemcpyAsync(p->d_cur_tables + (size_t)q * p->U, h, p->U * sizeof(void *), hipMemcpyHostToDevice, p->stream));
                        ptrs_uploaded = true;
                    }
                    HIP_TRY(scd::launch_sum_generic(p->d_cur_tables + (size_t)q * p->U, p->d_slot_table + pr.slot_off, p->d_slot_exp + pr.slot_off, (int)pr.tables.size(),
                                                    (int)pr.M, pairs_per_chunk, partials, grid, p->stream));
                }
            }
        }
        HIP_TRY(scd::launch_finalize(p->d_finprods, p->h_finprods.empty() ? nullptr : p->h_finprods.data(), p->d_W, (int)p->K, (int)p->D, grid, p->d_partials,
                                     p->d_scratch, p->d_chunk_msg, nullptr, nullptr, nullptr, 0, 1, p->d_fin_mb_counter, p->stream));
        const bool last = c + 1 == n_chunks;
        HIP_TRY(scd::launch_msg_accumulate(p->d_chunk_msg, p->d_chunk_msg + p->D, (int)p->D, c == 0, last, p->d_out, last ? d_wide : nullptr,
                                           (last && publish_to_host) ? p->h_out_dev : nullptr, (last && publish_to_host) ? p->h_flag_dev : nullptr, p->seq, p->stream));
        HIP_TRY(hipEventRecord(p->ev_consumed[q], p->stream));
    }
    if (bind) { // everything is resident now
        for (uint32_t u = 0; u < p->U; ++u) {
            Table &t = p->tabs[u];
            bool referenced = false;
            for (const Product &pr : p->prods)
                for (uint32_t tt : pr.tables) referenced |= tt == u;
            t.cur = t.buf[0];
            t.cur_top = (merged && p->use_f29 && referenced) ? t.buf_top[0] : nullptr;
            t.next = 1;
        }
    }
    p->timed = false;
    p->timing_pending = false;
    return SC_OK;
}

// deferred = true (library-internal): the challenge does not exist yet.  The round is enqueued behind a wait on p->sig and its
// bind kernel reads the challenge from the mailbox; provide_challenge() supplies it later.  Late (small) rounds only.
// SC_HOST_TRACE: report any single HIP call of a round's launch sequence that takes longer than a millisecond (stderr)
struct SlowCallProbe {
    const char *what;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit SlowCallProbe(const char *w) : what(w), on(std::getenv("SC_HOST_TRACE") != nullptr) {
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~SlowCallProbe() {
        if (!on) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 1.0) std::fprintf(stderr, "[sc] slow host call: %s took %.1f ms\n", what, ms);
    }
};
// ---- staged initialisation: IPForMLSumcheck::prover_init's deep copy (prover.rs:55-59) and the first prove_round in one pass -------------
// A Rust caller's tables are host memory: prover_init is a host-to-device copy of U 2^nv 32 bytes (5 GiB for config 3: ~95 ms over PCIe
// against 5 ms of proving).  Round 1 needs no challenge and its sums are additive over any split of the index range, so the copy goes in
// chunks on a second stream and k_round1_tree_split runs on chunk c while chunk c + 1 is in flight: each launch fills its own section of
// the node rows of the partial sums (RoundArgs::part_stride / part_block0), ONE finalize over all sections publishes the message under
// the sequence number round 1 will have, and keeps the node sums for round 2's claim identity -- exactly what an ordinary round 1 leaves.
// The tables are resident and the caller's memory is no longer referenced when sc_prover_init returns, as the reference's ownership has
// it; the first sc_prove_round finds its message waiting.  Shapes of the merged big-round kernel with at least 2^18 entries per table.
bool fin_mb_enabled();
bool staged_init_applies(const sc_prover *p) {
    return scd::policy(scd::kPolStagedInit) != 0 && p->merge_rounds && !p->any_generic && !p->streamed && !p->borrow && !p->fused_finalize && p->kernel_variant == 3 &&
           p->nv >= 18 && p->K > 0 && p->K <= (uint32_t)scd::kMetaProds;
}
int staged_copy_and_round1(sc_prover *p, const uint64_t *const *host_tables) {
    HIP_TRY(hipSetDevice(p->device));
    if (!p->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    if (!p->copy_stream2) HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream2, hipStreamNonBlocking));
    for (int q = 0; q < 2; ++q) {
        if (!p->ev_copied[q]) HIP_TRY(hipEventCreateWithFlags(&p->ev_copied[q], hipEventDisableTiming));
        if (!p->ev_copied2[q]) HIP_TRY(hipEventCreateWithFlags(&p->ev_copied2[q], hipEventDisableTiming));
    }
    const uint64_t n = 1ULL << p->nv, n_pairs = n >> 1;
    // the grid an ordinary round 1 of this shape takes (launch_round), cut into one section per chunk
    int G = n_pairs >= (1ULL << 21) ? 1024 : n_pairs >= (1ULL << 20) ? 768 : n_pairs >= (1ULL << 18) ? 384 : n_pairs >= (1ULL << 17) ? 256 : 192;
    if (p->K == 1) G = std::min(G, scd::kRoundTreeGrid);
    G = std::min(G, scd::grid_for_pairs(n_pairs));
    // Chunks of a half, a quarter, ... and the last two of equal size (1/32 of the tables from 2^21 entries, a quarter below): what is
    // left of round 1 when the last byte has arrived is the kernel over the LAST chunk alone.  The sections of the grid are not in
    // proportion: the early chunks' kernels hide under the copy whatever their grid, the late ones get an eighth of the blocks each so
    // that a small chunk still fills the chip (measured with proportional sections: 0.21 ms for 1/16 of config 3 on 64 blocks a product,
    // against 0.08 ms of work; profiles/r6e_staged_init_timeline.txt).
    // The copies alternate between two streams by table: a transfer costs the copy engine some 35 us of start-up and turn-around
    // (60 copies: 2 ms on one stream, against 94 ms of bytes), which a second transfer in flight hides.
    int levels = p->nv >= 21 ? 5 : 2;
    int sec[6], n_chunks = levels + 1;
    if (levels == 5 && G % 8 == 0) {
        sec[0] = G / 4, sec[1] = G / 8, sec[2] = G / 8;
        sec[3] = G / 8, sec[4] = G / 8, sec[5] = G / 4; // (the chunk the proof waits for gets a quarter)
    } else {
        levels = 2;
        while (levels > 0 && (G % (1 << levels)) != 0) --levels;
        n_chunks = levels + 1;
        for (int c = 0; c < n_chunks; ++c) sec[c] = G >> (c < levels ? c + 1 : levels);
    }
    for (uint32_t u = 0; u < p->U; ++u) // (before the first copy is enqueued: no return path leaves transfers from the caller's memory in flight)
        if (!host_tables[u]) return sc_internal_fail(SC_ERR_BAD_ARG, "table %u is null", u);
    scd::BindConst rc;
    std::memset(&rc, 0, sizeof(rc));
    // (whatever the handle's stream still holds -- a previous proof's last kernels read the buffers the copy overwrites -- goes first)
    HIP_TRY(hipEventRecord(p->ev_copied[1], p->stream));
    HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->ev_copied[1], 0));
    HIP_TRY(hipStreamWaitEvent(p->copy_stream2, p->ev_copied[1], 0));
    int block0 = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const int sh = c < levels ? c + 1 : levels;
        const uint64_t chunk = n >> sh, first = n - (n >> c); // entries per table in this chunk, entries before it
        const int gc = sec[c];                                // its section of the grid
        for (uint32_t u = 0; u < p->U; ++u) {
            HIP_TRY(hipMemcpyAsync(reinterpret_cast<char *>(p->tabs[u].buf[0]) + (size_t)first * 32, reinterpret_cast<const char *>(host_tables[u]) + (size_t)first * 32,
                                   (size_t)chunk * 32, hipMemcpyHostToDevice, (u & 1) ? p->copy_stream2 : p->copy_stream));
        }
        HIP_TRY(hipEventRecord(p->ev_copied[c & 1], p->copy_stream));
        HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_copied[c & 1], 0));
        if (p->U > 1) {
            HIP_TRY(hipEventRecord(p->ev_copied2[c & 1], p->copy_stream2));
            HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_copied2[c & 1], 0));
        }
        scd::RoundArgs ra;
        std::memset(&ra, 0, sizeof(ra));
        ra.n_prod = (int)p->K;
        for (uint32_t k = 0; k < p->K; ++k) {
            const Product &pr = p->prods[k];
            scd::TreeProd &tp = ra.prod[k];
            tp.M = pr.M;
            tp.partial_off = pr.partial_off;
            int f = 0;
            for (size_t s2 = 0; s2 < pr.tables.size(); ++s2)
                for (uint32_t rep = 0; rep < pr.exps[s2]; ++rep, ++f) {
                    tp.slot[f].exp = 1;
                    tp.slot[f].mode = 0;
                    tp.slot[f].src = p->tabs[pr.tables[s2]].buf[0] + 2 * (size_t)first; // (an element is two uint4)
                }
        }
        ra.part_stride = (uint32_t)G;
        ra.part_block0 = (uint32_t)block0;
        HIP_TRY(scd::launch_round_tree(ra, rc, chunk >> 1, p->d_partials, gc, p->stream, true, false));
        block0 += gc;
    }
    if (block0 != G) return sc_internal_fail(SC_ERR_HIP, "staged initialisation: the sections do not cover the grid");
    scd::plan_hit(scd::kPlanBigStagedRound1);
    const bool keeps = scd::finalize_keeps_sums((int)p->K, (int)p->D, G, !p->h_finprods.empty(), fin_mb_enabled());
    HIP_TRY(scd::launch_finalize(p->d_finprods, p->h_finprods.data(), p->d_W, (int)p->K, (int)p->D, G, p->d_partials, keeps ? p->d_sums[1] : p->d_scratch, p->d_out, nullptr,
                                 p->h_out_dev, p->h_flag_dev, p->seq + 1, 1, fin_mb_enabled() ? p->d_fin_mb_counter : nullptr, p->stream, nullptr));
    p->r1_cached = true;
    p->r1_keeps = keeps;
    return SC_OK;
}

int launch_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *d_wide, bool publish_to_host, bool deferred) {
    if (p->res.active) { // (sc_prove_round_partial after interactive rounds)
        int rc_q = resident_quiesce(p);
        if (rc_q) return rc_q;
    }
    DeviceGate gate(p->device);
    if (p->streamed && p->round < 2 && !p->exhausted) { // the inputs are still in host memory: the round is computed chunk by chunk
        if (deferred) return sc_internal_fail(SC_ERR_BAD_ARG, "streamed tables: rounds 1 and 2 are not pipelined");
        return launch_round_streamed(p, r_or_null, d_wide, publish_to_host);
    }
    // validation, same precedence as the reference's panics (prover.rs:78-98)
    if (p->exhausted) return sc_internal_fail(SC_ERR_NOT_ACTIVE, "Prover is not active");
    if (p->deferred_pending) return sc_internal_fail(SC_ERR_BAD_ARG, "a pipelined round is waiting for its challenge");
    if (r_or_null && p->round == 0) return sc_internal_fail(SC_ERR_FIRST_ROUND_HAS_MSG, "first round should be prover first.");
    if (!r_or_null && p->round > 0 && !deferred) return sc_internal_fail(SC_ERR_MISSING_MSG, "verifier message is empty");
    if (p->round + 1 > p->nv) return sc_internal_fail(SC_ERR_NOT_ACTIVE, "Prover is not active");
    sch::Fr r = sch::zero();
    if (r_or_null) {
        std::memcpy(&r, r_or_null, 32);
        if (sch::geq_p(r)) return sc_internal_fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
    }
    if (deferred) { // (every check comes before the first change to the handle)
        const uint64_t np = 1ULL << (p->nv - (p->round + 1));
        if (!(np <= small_pairs_limit() && p->U <= (uint32_t)scd::kMaxSmallTables && p->K > 0)) return sc_internal_fail(SC_ERR_BAD_ARG, "only late rounds are pipelined");
    }
    HIP_TRY(hipSetDevice(p->device));
    if (!deferred) { // (a pipelined round records no events: collecting would wait for the round before it)
        int rc_t = collect_timing(p);
        if (rc_t) return rc_t;
    }
    if (p->r1_cached) { // a staged initialisation computed this round under the copy and published it under the next sequence number
        p->r1_cached = false;
        if (p->round == 0 && !r_or_null && !deferred && !d_wide && publish_to_host) {
            p->round = 1;
            p->seq += 1;
            p->sums_round = p->r1_keeps ? 1 : -1;
            p->timed = false;
            p->timing_pending = false;
            p->prod_timed = false;
            return SC_OK;
        }
        // (a caller that wants the lanes of a sharded round instead: the round is computed again, over the resident tables)
    }
    bool bind = r_or_null != nullptr || deferred;
    if (r_or_null) p->randomness.push_back(r);
    p->round += 1;
    const uint64_t n_pairs = 1ULL << (p->nv - p->round);
    const FrHost rdev = to_dev(r);
    sch::Fr r32v = r; // r * 2^5 for the 2^261-radix kernels
    for (int d = 0; d < 5; ++d) r32v = sch::add(r32v, r32v);
    const FrHost r32 = to_dev(r32v);
    int scaled = 0;
    const uint64_t small_pairs = small_pairs_limit();
    const bool small_round = n_pairs <= small_pairs && p->K > 0 && (p->U <= (uint32_t)scd::kMaxSmallTables || p->d_cur_tables != nullptr);
#ifdef SC_EXPERIMENTS
    const bool tiled = !small_round && !p->any_generic && p->kernel_variant == 2;
#else
    const bool tiled = false;
    (void)tiled;
#endif
    scd::BindConst rc; // (only the big rounds of the tree kernels pay for it)
    std::memset(&rc, 0, sizeof(rc)); // tree kernels: rows (r * 2^(29 i + 58)) mod p as plain 29-bit limbs (fe_device.hpp, fe_mul_bind)
    if (bind && !small_round && p->kernel_variant == 3) make_bind_const(r, rc);
    int grid = scd::grid_for_pairs(n_pairs);
#ifdef SC_EXPERIMENTS
    if (tiled) grid = scd::grid_for_tiles(n_pairs);
#endif
    const bool timed = p->timing && !deferred;
    if (timed) HIP_TRY(hipEventRecord(p->ev0, p->stream));
    const FrHost *r_mail = nullptr;
    if (deferred) {
        p->sig_seq += 1;
        {
            SlowCallProbe pr("launch k_wait_challenge");
            HIP_TRY(scd::launch_wait_challenge(p->sig_dev, p->sig_seq, p->h_mail_dev + (p->sig_seq & 1u), p->d_mail + (p->sig_seq & 1u), p->stream));
        }
        r_mail = p->d_mail + (p->sig_seq & 1u);
        p->deferred_pending = true;
    }

    auto bind_table = [&](uint32_t u) -> hipError_t { // stand-alone bind of table u (2*n_pairs outputs)
        Table &t = p->tabs[u];
        uint4 *dst = t.buf[t.next];
        hipError_t e = scd::launch_fix(t.cur, dst, rdev, 2 * n_pairs, p->stream);
        t.cur = dst;
        t.next ^= 1;
        return e;
    };

    const bool small = small_round;
    if (small) {
        // latency-bound round: one launch binds every table (32 tables a launch), one launch sums every (product, point) combination
        TablePtrs tp;
        if (bind) {
            for (uint32_t u0 = 0; u0 < p->U; u0 += (uint32_t)scd::kMaxSmallTables) {
                const uint32_t cnt = std::min<uint32_t>(p->U - u0, (uint32_t)scd::kMaxSmallTables);
                std::memset(&tp, 0, sizeof(tp));
                for (uint32_t j = 0; j < cnt; ++j) {
                    Table &t = p->tabs[u0 + j];
                    tp.src[j] = t.cur;
                    tp.src_top[j] = t.cur_top;
                    tp.dst[j] = t.buf[t.next];
                }
                {
                    SlowCallProbe pr("launch k_fix_multi");
                    HIP_TRY(scd::launch_fix_multi(tp, (int)cnt, rdev, r_mail, 2 * n_pairs, p->stream));
                }
                for (uint32_t j = 0; j < cnt; ++j) {
                    Table &t = p->tabs[u0 + j];
                    t.cur = t.buf[t.next];
                    t.cur_top = nullptr; // the latency-bound path keeps tables canonical in the reference layout
                    t.next ^= 1;
                }
            }
        }
        SlowCallProbe pr_sum("launch k_sum_combos");
        scd::plan_hit(deferred ? scd::kPlanSmallPipelined : p->U > (uint32_t)scd::kMaxSmallTables ? scd::kPlanSmallPtrs : p->has_meta ? scd::kPlanSmallLaunched : scd::kPlanSmallCombosTable);
        if (p->U <= (uint32_t)scd::kMaxSmallTables) {
            std::memset(&tp, 0, sizeof(tp));
            for (uint32_t u = 0; u < p->U; ++u) tp.src[u] = p->tabs[u].cur;
            if (p->has_meta) HIP_TRY(scd::launch_sum_combos_meta(tp, p->meta, p->n_combos, n_pairs, p->d_partials, grid, p->stream));
            else HIP_TRY(scd::launch_sum_combos(tp, p->d_combos, p->n_combos, p->d_slot_table, p->d_slot_exp, n_pairs, p->d_partials, grid, p->stream));
        } else { // more tables than a launch's arguments hold: the pointers go through device memory (never a pipelined round: can_defer_next)
            for (uint32_t u = 0; u < p->U; ++u) p->h_cur_tables[u] = p->tabs[u].cur;
            HIP_TRY(hipMemcpyAsync(p->d_cur_tables, p->h_cur_tables, p->U * sizeof(void *), hipMemcpyHostToDevice, p->stream));
            HIP_TRY(scd::launch_sum_combos_ptrs(p->d_cur_tables, p->d_combos, p->n_combos, p->d_slot_table, p->d_slot_exp, n_pairs, p->d_partials, grid, p->stream));
        }
        scaled = 1; // products of up to kMaxFusedM multiplicands are summed in carry-free arithmetic (2^261 radix) there too
        bind = false;
    }
    if (bind && p->any_generic) { // products beyond kMaxFusedM read bound tables: bind everything up front, 32 tables a launch
        scd::plan_hit(scd::kPlanBigBindPass);
        for (uint32_t u0 = 0; u0 < p->U; u0 += (uint32_t)scd::kMaxSmallTables) {
            const uint32_t cnt = std::min<uint32_t>(p->U - u0, (uint32_t)scd::kMaxSmallTables);
            TablePtrs tp;
            std::memset(&tp, 0, sizeof(tp));
            for (uint32_t j = 0; j < cnt; ++j) {
                Table &t = p->tabs[u0 + j];
                tp.src[j] = t.cur;
                tp.src_top[j] = t.cur_top;
                tp.dst[j] = t.buf[t.next];
            }
            HIP_TRY(scd::launch_fix_multi(tp, (int)cnt, rdev, nullptr, 2 * n_pairs, p->stream));
            for (uint32_t j = 0; j < cnt; ++j) {
                Table &t = p->tabs[u0 + j];
                t.cur = t.buf[t.next];
                t.cur_top = nullptr;
                t.next ^= 1;
            }
        }
        bind = false;
    }
    std::vector<uint8_t> bound(p->U, 0);
    bool ptrs_uploaded = false;
    bool finalized = false; // the merged big-round launch also produced the message
    bool skip1 = false;     // the round kernel leaves node 1 out (ClaimArgs)
    const bool merged = !small && p->merge_rounds && !p->any_generic;
    if (merged) {
        grid = std::min(grid, scd::kRoundTreeGrid);
        // One product per block row (k_round_tree_split / k_round1_tree_split): `grid` blocks per product.  Measured per round size on
        // config 3 (profiles/r2e_split_rounds.txt): many small blocks for the rounds that stream tables, fewer for the short ones.
        bool split = !p->fused_finalize;
        int split_grid = n_pairs >= (1ULL << 21) ? 1024 : n_pairs >= (1ULL << 20) ? 768 : n_pairs >= (1ULL << 18) ? 384 : n_pairs >= (1ULL << 17) ? 256 : 192;
#ifdef SC_EXPERIMENTS // SC_SPLIT=0: every product in every block (k_round_tree); SC_SPLIT_GRID=n: blocks per product
        static const bool split_off = std::getenv("SC_SPLIT") && std::atoi(std::getenv("SC_SPLIT")) == 0;
        static const int split_cap = std::getenv("SC_SPLIT_GRID") ? std::atoi(std::getenv("SC_SPLIT_GRID")) : 0;
        if (split_off) split = false;
        if (split_cap > 0) split_grid = split_cap;
#endif
        // The block counts above were measured on config 3's FOUR rows; what a short round needs is enough blocks in all to keep the per-lane
        // chain at one iteration.  Fewer rows get proportionally more blocks per row in the rounds that no longer stream (< 2^21 pairs);
        // the streaming rounds of a single row keep one full wave of resident blocks (a second, partial wave would run alone at the end).
#ifndef SC_NO_KGRID // (A/B build: the per-row counts whatever the number of rows)
        if (p->K < 4 && n_pairs < (1ULL << 21)) split_grid = std::min(scd::kMaxGrid, split_grid * 4 / (int)p->K);
        else if (p->K == 1) split_grid = std::min(split_grid, scd::kRoundTreeGrid); // (one product: one full wave of resident blocks)
#else
        if (p->K == 1) split_grid = std::min(split_grid, scd::kRoundTreeGrid);
#endif
        if (split) grid = std::min(scd::grid_for_pairs(n_pairs), split_grid);
        // the previous round's complete node sums are on the device and this round's will be: node 1 comes from the claim identity
        skip1 = bind && split && p->sums_round == (int64_t)p->round - 1 && skip1_enabled() &&
                scd::finalize_keeps_sums((int)p->K, (int)p->D, grid, !p->h_finprods.empty(), fin_mb_enabled());
        // One launch for the round.  The first factor touching a table binds and stores it (mode 1); every later factor on
        // that table -- in the same or in another product -- re-binds from the old buffer without storing (mode 3), so no
        // product reads what another one writes in this launch.
        scd::RoundArgs ra;
        std::memset(&ra, 0, sizeof(ra));
        ra.n_prod = (int)p->K;
        std::vector<const uint4 *> old_src(p->U);
        std::vector<const int32_t *> old_top(p->U);
        for (uint32_t u = 0; u < p->U; ++u) {
            old_src[u] = p->tabs[u].cur;
            old_top[u] = p->tabs[u].cur_top;
        }
        for (uint32_t k = 0; k < p->K; ++k) {
            const Product &pr = p->prods[k];
            scd::TreeProd &tp = ra.prod[k];
            tp.M = pr.M;
            tp.partial_off = pr.partial_off;
            int f = 0;
            for (size_t s = 0; s < pr.tables.size(); ++s) {
                const uint32_t u = pr.tables[s];
                Table &t = p->tabs[u];
                for (uint32_t rep = 0; rep < pr.exps[s]; ++rep, ++f) {
                    scd::Slot &sl = tp.slot[f];
                    sl.exp = 1;
                    sl.src = old_src[u];
                    sl.src_top = old_top[u];
                    if (!bind) {
                        sl.mode = 0;
                    } else if (!bound[u]) {
                        sl.mode = 1;
                        sl.dst = t.buf[t.next];
                        sl.dst_top = p->use_f29 ? t.buf_top[t.next] : nullptr;
                        t.cur = t.buf[t.next];
                        t.cur_top = sl.dst_top;
                        t.next ^= 1;
                        bound[u] = 1;
                    } else {
                        sl.mode = 3;
                        sl.dst_top = p->use_f29 ? t.buf_top[0] : nullptr; // only selects the carry-pass path
                    }
                }
            }
        }
        // the finalize step runs inside the launch (the blocks that finish last add up the partials and publish the message)
        p->seq += 1;
        ra.fin.enabled = p->fused_finalize ? 1 : 0;
        ra.fin.D = (int)p->D;
        for (uint32_t k = 0; k < p->K; ++k) ra.fin.w_off[k] = p->h_finprods[k].w_off;
        ra.fin.Wm = reinterpret_cast<const uint4 *>(p->d_W);
        ra.fin.partials2 = reinterpret_cast<uint4 *>(p->d_partials2);
        ra.fin.counters = p->d_fin_counters;
        ra.fin.out = reinterpret_cast<uint4 *>(p->d_out);
        ra.fin.out_wide = d_wide;
        ra.fin.h_out = publish_to_host ? reinterpret_cast<uint4 *>(p->h_out_dev) : nullptr;
        ra.fin.h_flag = publish_to_host ? p->h_flag_dev : nullptr;
        ra.fin.seq = p->seq;
        if (p->timing) HIP_TRY(hipEventRecord(p->prod_ev[0], p->stream));
        if (bind) scd::plan_hit(p->use_f29 ? scd::kPlanBigF29Store : scd::kPlanBigCanonicalStore);
        HIP_TRY(scd::launch_round_tree(ra, rc, n_pairs, p->d_partials, grid, p->stream, split, skip1));
        if (p->timing) HIP_TRY(hipEventRecord(p->prod_ev[1], p->stream));
        scaled = 1;
        if (p->fused_finalize) finalized = true;
        else p->seq -= 1;
    }
    for (uint32_t k = 0; k < p->K && !small && !merged; ++k) {
        const Product &pr = p->prods[k];
        FrHost *partials = p->d_partials + pr.partial_off;
        if (p->timing) HIP_TRY(hipEventRecord(p->prod_ev[2 * k], p->stream));
        if (pr.fused && p->kernel_variant == 3 && (pr.M <= 4 || p->wide_tree)) {
            // product tree: one argument slot per FACTOR.  The first factor touching a table this round binds and stores it;
            // a repeat inside the same product re-binds from the old table without storing (mode 3).
            ProdArgs a;
            std::memset(&a, 0, sizeof(a));
            a.n_slots = (int)pr.M;
            int f = 0;
            for (size_t s = 0; s < pr.tables.size(); ++s) {
                Table &t = p->tabs[pr.tables[s]];
                const uint4 *old_src = t.cur;
                const int32_t *old_top = t.cur_top;
                bool stored_here = false;
                for (uint32_t rep = 0; rep < pr.exps[s]; ++rep, ++f) {
                    a.slot[f].exp = 1;
                    if (bind && !bound[pr.tables[s]]) {
                        a.slot[f].mode = 1;
                        a.slot[f].src = old_src;
                        a.slot[f].src_top = old_top;
                        a.slot[f].dst = t.buf[t.next];
                        a.slot[f].dst_top = p->use_f29 ? t.buf_top[t.next] : nullptr;
                        t.cur = t.buf[t.next];
                        t.cur_top = a.slot[f].dst_top;
                        t.next ^= 1;
                        bound[pr.tables[s]] = 1;
                        stored_here = true;
                    } else if (stored_here) {
                        a.slot[f].mode = 3;
                        a.slot[f].src = old_src;
                        a.slot[f].src_top = old_top;
                        a.slot[f].dst = nullptr;
                        a.slot[f].dst_top = p->use_f29 ? t.buf_top[0] : nullptr; // only selects the tighten path
                    } else {
                        a.slot[f].mode = 0;
                        a.slot[f].src = t.cur;
                        a.slot[f].src_top = t.cur_top;
                        a.slot[f].dst = nullptr;
                    }
                }
            }
            scd::plan_hit(pr.M <= 4 ? scd::kPlanBigPerProductTree : scd::kPlanBigWide);
            if (bind) scd::plan_hit(p->use_f29 ? scd::kPlanBigF29Store : scd::kPlanBigCanonicalStore);
            HIP_TRY(scd::launch_prod_tree((int)pr.M, a, rc, n_pairs, partials, grid, p->stream));
            scaled = 1;
        } else if (pr.fused) {
            ProdArgs a;
            std::memset(&a, 0, sizeof(a));
            a.n_slots = (int)pr.tables.size();
            for (size_t s = 0; s < pr.tables.size(); ++s) {
                Table &t = p->tabs[pr.tables[s]];
                a.slot[s].exp = pr.exps[s];
                if (bind && !bound[pr.tables[s]]) { // first product touching this table this round binds it
                    a.slot[s].mode = 1;
                    a.slot[s].src = t.cur;
                    a.slot[s].dst = t.buf[t.next];
                    t.cur = t.buf[t.next];
                    t.next ^= 1;
                    bound[pr.tables[s]] = 1;
                } else {
                    a.slot[s].mode = 0;
                    a.slot[s].src = t.cur;
                    a.slot[s].dst = nullptr;
                }
            }
#ifdef SC_EXPERIMENTS
            if (tiled) {
                HIP_TRY(scd::launch_round_tile((int)pr.M, a, r32, n_pairs, partials, grid, p->stream));
                scaled = 1;
            } else if (!p->use_fe) {
                HIP_TRY(scd::launch_prod_round((int)pr.M, a, rdev, n_pairs, partials, grid, p->stream));
            } else
#endif
            {
                scd::plan_hit(scd::kPlanBigNodeByNode);
                HIP_TRY(scd::launch_prod_round_fe((int)pr.M, a, r32, n_pairs, partials, grid, p->stream));
                scaled = 1;
            }
        } else if (pr.M <= (uint32_t)scd::kMaxWideM && p->wide_tree && p->kernel_variant == 3) {
            // nine to twelve multiplicands: a tree of the trees (kernels_wide16.hip) over the tables the bind pass above left; one slot per
            // FACTOR, and the sums come back in k_sum_generic's form (the kernel takes its 2^(-5(M-1)) off again)
            scd::WideArgs16 a;
            std::memset(&a, 0, sizeof(a));
            const std::vector<uint32_t> &factors = p->prod_indices[k];
            a.n_slots = (int)factors.size();
            for (size_t f = 0; f < factors.size(); ++f) {
                const Table &t = p->tabs[factors[f]];
                a.slot[f].mode = 0;
                a.slot[f].exp = 1;
                a.slot[f].src = t.cur;
                a.slot[f].src_top = t.cur_top;
            }
            sch::Fr comp = sch::kOne; // 2^(5(M-1)) in Montgomery form
            for (uint32_t dbl = 0; dbl < 5 * (pr.M - 1); ++dbl) comp = sch::add(comp, comp);
            scd::plan_hit(scd::kPlanBigWide16);
            HIP_TRY(scd::launch_prod_tree_wide16((int)pr.M, a, to_dev(comp), n_pairs, partials, grid, p->stream));
        } else {
            if (!ptrs_uploaded) {
                for (uint32_t u = 0; u < p->U; ++u) p->h_cur_tables[u] = p->tabs[u].cur;
                HIP_TRY(hipMemcpyAsync(p->d_cur_tables, p->h_cur_tables, p->U * sizeof(void *), hipMemcpyHostToDevice, p->stream));
                ptrs_uploaded = true;
            }
            scd::plan_hit(scd::kPlanBigGeneric);
            HIP_TRY(scd::launch_sum_generic(p->d_cur_tables, p->d_slot_table + pr.slot_off, p->d_slot_exp + pr.slot_off,
                                            (int)pr.tables.size(), (int)pr.M, n_pairs, partials, grid, p->stream));
        }
        if (p->timing) HIP_TRY(hipEventRecord(p->prod_ev[2 * k + 1], p->stream));
    }
    if (bind) { // tables that no product refers to still follow the state machine
        for (uint32_t u = 0; u < p->U; ++u)
            if (!bound[u]) HIP_TRY(bind_table(u));
    }
    if (!finalized) {
    p->seq += 1;
    // the multi-block form leaves the round's node sums behind: kept per round parity for the next round's claims (tree rounds only:
    // their sums all carry the same scaling)
    const bool keeps = scd::finalize_keeps_sums((int)p->K, (int)p->D, grid, !p->h_finprods.empty(), fin_mb_enabled());
    scd::ClaimArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    if (skip1) {
        ca.skip1 = 1;
        ca.prev = reinterpret_cast<const uint4 *>(p->d_sums[(p->round - 1) & 1]);
        bool done[5] = {false, false, false, false, false};
        for (uint32_t k = 0; k < p->K; ++k) {
            const uint32_t M = p->prods[k].M;
            if (done[M]) continue;
            done[M] = true;
            sch::Fr lam[5];
            claim_weights(M, r, lam);
            for (uint32_t s2 = 0; s2 <= M; ++s2) ca.lam[scd::claim_off((int)M) + (int)s2] = to_dev(lam[s2]);
        }
    }
    SlowCallProbe pr_fin("launch k_finalize");
    HIP_TRY(scd::launch_finalize(p->d_finprods, p->h_finprods.empty() ? nullptr : p->h_finprods.data(), p->d_W, (int)p->K, (int)p->D, grid, p->d_partials,
                                 keeps ? p->d_sums[p->round & 1] : p->d_scratch, p->d_out, d_wide,
                                 publish_to_host ? p->h_out_dev : nullptr, publish_to_host ? p->h_flag_dev : nullptr,
                                 (p->wide_tagged && d_wide) ? p->wide_gen : p->seq, // (tagged lanes: the communicator's generation, the same on every rank)
                                 scaled | ((p->wide_tagged && d_wide) ? 2 : 0), fin_mb_enabled() ? p->d_fin_mb_counter : nullptr, p->stream, skip1 ? &ca : nullptr));
    p->sums_round = keeps && merged ? (int64_t)p->round : -1;
    }
    if (timed) HIP_TRY(hipEventRecord(p->ev1, p->stream));
    if (!deferred) { // (a pipelined round leaves the previous round's pending event pairs to the next collect_timing)
        p->timed = timed;
        p->timing_pending = timed;
        p->prod_timed = timed && !small;
        p->prod_merged = merged;
        p->timed_round = p->round;
    }
    return SC_OK;
}

int await_round(sc_prover *p, uint64_t *out_evals, uint32_t want);

int resident_start(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals);
int resident_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals);
extern "C" int sc_prove_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals) {
    if (!p || !out_evals) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    // late rounds of the interactive protocol: a kernel that stays on the GPU between calls (see resident_start)
    int rc = p->res.active ? resident_round(p, r_or_null, out_evals) : resident_start(p, r_or_null, out_evals);
    if (rc != kResidentGone) return rc;
    rc = launch_round(p, r_or_null, nullptr, true);
    if (rc) return rc;
    return await_round(p, out_evals, p->seq);
}

// want: the sequence number the awaited round's finalize publishes (p->seq right after that round was launched)
int await_round(sc_prover *p, uint64_t *out_evals, const uint32_t want) {
    // The message is written by k_finalize straight into host-mapped pinned memory, followed by a system-scope release of
    // the sequence flag: poll it instead of paying a DMA copy plus an interrupt-driven stream synchronise every round.
    uint64_t spins = 0;
    bool seen = false;
    const auto t_start = std::chrono::steady_clock::now();
    while (!(seen = (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == want))) {
        if ((++spins & 0xfff) == 0) {
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) break; // fall back to a real sync
        }
    }
    if (!seen) {
        if (p->deferred_pending) { // the stream cannot be synchronised while the next round waits for its challenge
            abandon_deferred(p);
            return sc_internal_fail(SC_ERR_HIP, "round did not publish its message within 2 s");
        }
        {
            DeviceGate gate_(p->device);
            HIP_TRY(hipStreamSynchronize(p->stream));
        }
        if (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) != want) return sc_internal_fail(SC_ERR_HIP, "round finished without publishing its message");
    }
    if (wait_gave_up(p)) { // a wait kernel's bound expired before its challenge arrived: that round ran on a stale one
        if (std::getenv("SC_HOST_TRACE"))
            std::fprintf(stderr, "[sc] give-up seen in await_round: marker %u, sig word %u, sig_seq %u, awaited seq %u, h_flag %u, round %u, deferred_pending %d, waited %.3f s\n",
                         __atomic_load_n(p->sig + 1, __ATOMIC_ACQUIRE), __atomic_load_n(p->sig, __ATOMIC_ACQUIRE), p->sig_seq, want,
                         __atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE), p->round, (int)p->deferred_pending,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        abandon_deferred(p);
        p->exhausted = true;
        return sc_internal_fail(SC_ERR_HIP, "the host took longer than the wait kernel's bound to deliver a challenge; the proof is void");
    }
    std::memcpy(out_evals, p->h_out, (size_t)p->D * 32);
    return SC_OK;
}

// ---- the persistent tail: every remaining latency-bound round in ONE kernel launch (kernels.hip: k_tail_rounds) -------------
// Usable when the round metadata fits kernel arguments (tail_shape_ok), the next round is a small one, and launches are
// asynchronous (the kernel waits for the host; sc_set_policy("pipeline", 0) switches it off together with the pipelined rounds).
constexpr size_t kTailSyncBytes = 4 * (16 + (size_t)scd::kTailMaxGrid);
// ONE tail kernel per device at a time: its grid barrier needs every launched block resident, and the grid is sized for an otherwise
// idle GPU (tail_max_resident_blocks); two of them from two proving threads could each end up partially resident and wait for
// blocks that are never scheduled.  A prover that finds the slot taken does not wait for it: its late rounds run as pipelined
// launches (the path every proof took before the tail kernel existed).  (The kernel's waits are bounded as well: kernels.hip, grid_barrier.)
// The slot has an OWNER (a handle), taken and given back under a per-device mutex.  A resident kernel of the interactive protocol holds
// it for as long as the kernel may be on the GPU -- but its patience is ~0.5 ms, while the handle may sit idle mid-protocol for as long as
// its verifier likes and only notices that its kernel left on its next call.  So a slot whose holder is a resident kernel that has
// raised its exit marker (sig[1], host-mapped: the kernel's last store before every block returns) counts as free: the next prover
// takes it over, and the former holder's release becomes a no-op.
// process-wide counters a host can read (sc_library_stats): which path the late rounds took, what was retried
std::atomic<uint64_t> g_stat[8];
struct TailOwner {
    std::mutex mu;
    sc_prover *owner = nullptr;
    bool resident = false;           // held by resident_start (reclaimable once the kernel has left)
    const uint32_t *marker = nullptr; // the holder's sig + 1
};
static TailOwner g_tail_owner[64];
bool tail_slot_acquire(sc_prover *p, bool resident) {
    TailOwner &t = g_tail_owner[(unsigned)p->device & 63u];
    std::lock_guard<std::mutex> lk(t.mu);
    if (t.owner && t.owner != p) {
        if (!(t.resident && t.marker && __atomic_load_n(t.marker, __ATOMIC_ACQUIRE) != 0)) {
            g_stat[kStatTailSlotBusy].fetch_add(1, std::memory_order_relaxed);
            return false;
        }
        g_stat[kStatTailSlotReclaims].fetch_add(1, std::memory_order_relaxed);
    }
    t.owner = p;
    t.resident = resident;
    t.marker = p->sig ? p->sig + 1 : nullptr;
    return true;
}
void tail_slot_release(sc_prover *p) { // (a holder that lost the slot to a reclaim releases nothing)
    TailOwner &t = g_tail_owner[(unsigned)p->device & 63u];
    std::lock_guard<std::mutex> lk(t.mu);
    if (t.owner == p) {
        t.owner = nullptr;
        t.marker = nullptr;
    }
}
struct TailSlot {
    sc_prover *const p;
    bool held;
    explicit TailSlot(sc_prover *p_) : p(p_), held(tail_slot_acquire(p_, false)) {}
    ~TailSlot() {
        if (held) tail_slot_release(p);
    }
    TailSlot(const TailSlot &) = delete;
    TailSlot &operator=(const TailSlot &) = delete;
};
bool tail_shape_ok(const sc_prover *p) {
    return p->use_tail && p->K > 0 && p->U <= (uint32_t)scd::kMaxSmallTables && p->has_meta && p->K <= (uint32_t)scd::kMetaProds &&
           (size_t)p->K * p->D * (p->D + 2) * 32 <= 48 * 1024;
}
int tail_slices_blocks_for(sc_prover *p);
// `slices`: the caller runs every remaining round (sc_ml_prove*, GKR): where the tables fit LDS as slices (k_tail_slices) the persistent
// kernel takes over from the first latency-bound round (<= 2^14 pairs); k_tail_rounds -- the interactive protocol's resident kernel, and
// shapes whose slices do not fit -- from 2048 pairs
bool tail_possible(sc_prover *p, bool slices = false) {
    if (!tail_shape_ok(p) || p->exhausted || p->round >= p->nv || p->deferred_pending) return false;
    if (p->streamed && p->round < 2) return false;
    const uint64_t next_pairs = 1ULL << (p->nv - (p->round + 1));
    if (next_pairs > scd::kTailMaxPairs && !(slices && next_pairs <= scd::kTsMaxPairs && tail_slices_blocks_for(p) > 0)) return false;
    if (!ensure_mailbox(p)) return false;
    if (!p->d_tail_sync) {
        // 16 sync words + one arrival flag per block | 2 challenge slots | K * D node sums
        if (hipMalloc(reinterpret_cast<void **>(&p->d_tail_sync), kTailSyncBytes + 64 + (size_t)p->K * p->D * 32) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        p->tail_max_blocks = scd::tail_max_resident_blocks(p->device);
    }
    return p->tail_max_blocks > 0;
}

// Launch k_tail_rounds for the handle's next n_rounds rounds (the caller holds the device gate and the device's tail slot);
// r_or_null = the challenge the first of them binds; max_spins = how long block 0 waits for each later challenge.
// The tables resident in LDS for the whole tail (kernels_tail.hip: k_tail_slices) where the shape allows: every product in carry-free
// arithmetic, the slices within a CU's LDS.  sc_set_policy("tail_slices", 0): k_tail_rounds everywhere (A/B runs, tests of the older path).
int tail_slices_blocks_for(sc_prover *p) {
    if (scd::policy(scd::kPolTailSlices) == 0 || p->max_mult > (uint32_t)(p->wide_tree ? scd::kMaxWideM : scd::kMaxFusedM) || p->round >= p->nv) return 0;
    const int B = scd::tail_slices_blocks(1ULL << (p->nv - (p->round + 1)), (int)p->U, (int)p->K, (int)p->D, p->n_combos, (int)p->max_mult,
                                          scd::tail_slices_max_blocks(p->device, (int)p->max_mult));
    if (B <= 0) return 0;
    if (!p->d_tail_xw) { // tagged hand-over words, owned by the handle: zero once, tags only ever grow
        if (hipMalloc(reinterpret_cast<void **>(&p->d_tail_xw), scd::kTsXwWords * 8) != hipSuccess ||
            hipMemsetAsync(p->d_tail_xw, 0, scd::kTsXwWords * 8, p->stream) != hipSuccess) {
            (void)hipGetLastError();
            if (p->d_tail_xw) (void)hipFree(p->d_tail_xw);
            p->d_tail_xw = nullptr;
            return 0;
        }
        p->ts_tag = 1;
        // the mailbox in device memory, where the host may store into it (sc_set_policy("vram_mailbox", 0): block 0 polls the host-mapped one and passes it on)
        int large_bar = 0;
        if (scd::policy(scd::kPolVramMailbox) != 0 && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, p->device) == hipSuccess && large_bar) {
            void *m = nullptr;
            if (hipExtMallocWithFlags(&m, 256, hipDeviceMallocFinegrained) == hipSuccess && hipMemsetAsync(m, 0, 256, p->stream) == hipSuccess &&
                hipStreamSynchronize(p->stream) == hipSuccess)
                p->d_vram_mail = static_cast<uint64_t *>(m);
            else if (m)
                (void)hipFree(m);
        }
        (void)hipGetLastError();
    }
    return B;
}
int tail_launch(sc_prover *p, uint32_t n_rounds, const sch::Fr *r_or_null, uint32_t max_spins, scd::TailArgs &A, int &grid, int slices_B = 0, bool host_mailbox_only = false) {
    const uint32_t D = p->D;
    std::memset(&A, 0, sizeof(A));
    for (uint32_t u = 0; u < p->U; ++u) {
        Table &t = p->tabs[u];
        A.t.cur0[u] = t.cur;
        A.t.cur0_top[u] = t.cur_top;
        A.t.b0[u] = t.buf[t.next];
        A.t.b1[u] = t.buf[t.next ^ 1];
    }
    A.n_tables = (int)p->U;
    A.n_rounds = (int)n_rounds;
    A.first_has_bind = r_or_null ? 1 : 0;
    A.first_pairs = 1ULL << (p->nv - (p->round + 1));
    if (r_or_null) A.r0 = to_dev(*r_or_null);
    A.n_combos = p->n_combos;
    A.K = (int)p->K;
    A.D = (int)D;
    A.Wm = reinterpret_cast<const uint4 *>(p->d_W);
    A.partials = reinterpret_cast<uint4 *>(p->d_partials);
    A.sync = p->d_tail_sync;
    A.chal = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(p->d_tail_sync) + kTailSyncBytes);
    A.sums = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(p->d_tail_sync) + kTailSyncBytes + 64);
    A.h_out = reinterpret_cast<uint4 *>(p->h_out_dev);
    A.h_flag = p->h_flag_dev;
    A.seq0 = p->seq + 1;
    A.sig = p->sig_dev;
    A.mail_host = reinterpret_cast<const uint64_t *>(p->h_mail_dev) + 16; // the tagged slots (byte offset 128)
    A.sig0 = p->sig_seq;
    A.max_spins = max_spins;
    scd::FinMeta fm;
    std::memset(&fm, 0, sizeof(fm));
    std::memcpy(fm.prod, p->h_finprods.data(), (size_t)p->K * sizeof(FinProd));
    grid = 1; // (the kernel's tail_active_blocks for the first round)
    if (A.first_pairs > scd::tail_flat_pairs(p->n_combos)) {
        const uint64_t bind_blocks = (2 * A.first_pairs * p->U + scd::kBlock - 1) / scd::kBlock;
        const uint64_t sum_blocks = ((A.first_pairs + scd::kBlock - 1) / scd::kBlock) * (uint64_t)p->n_combos;
        grid = (int)std::min<uint64_t>((uint64_t)p->tail_max_blocks, std::max(bind_blocks, sum_blocks));
    }
    if (slices_B > 0) {
        HIP_TRY(scd::launch_zero_words(p->d_tail_sync, (uint32_t)((kTailSyncBytes + 64) / 4), p->stream, reinterpret_cast<uint32_t *>(p->d_tail_xw),
                                       (uint32_t)(scd::kTsAccWords * 2))); // ... and the accumulator ring, in the same launch
        scd::TailSlicesArgs S;
        std::memset(&S, 0, sizeof(S));
        S.base = A;
        S.B = slices_B;
        S.xw = p->d_tail_xw;
        S.tag0 = p->ts_tag;
        S.mail_vram = host_mailbox_only ? nullptr : p->d_vram_mail;
        p->ts_tag += n_rounds;
        grid = slices_B;
        HIP_TRY(scd::launch_tail_slices(S, p->meta, fm, (int)p->max_mult, p->stream));
        return SC_OK;
    }
    HIP_TRY(scd::launch_zero_words(p->d_tail_sync, (uint32_t)((kTailSyncBytes + 64) / 4), p->stream));
    HIP_TRY(scd::launch_tail_rounds(A, p->meta, fm, grid, p->stream));
    return SC_OK;
}
// where the tables are after a tail kernel that did `nb` binds
void tail_epilogue_tables(sc_prover *p, uint32_t nb) {
    if (nb == 0) return;
    for (uint32_t u = 0; u < p->U; ++u) {
        Table &t = p->tabs[u];
        uint4 *b0 = t.buf[t.next], *b1 = t.buf[t.next ^ 1];
        t.cur = (nb & 1) ? b0 : b1;
        t.cur_top = nullptr;
        if (nb & 1) t.next ^= 1;
    }
}
// the host writes challenge `vm` for the poll that waits for tag `sv`: 32-bit limb i, tagged -- every word validates itself, the
// device's poll IS the fetch
void tail_post_challenge(sc_prover *p, uint32_t sv, const sch::Fr &vm, bool vram = false) {
    uint64_t *slot = reinterpret_cast<uint64_t *>(p->h_mail) + 16 + 8 * (sv & 1u);
    if (vram) {
        // k_tail_slices with a mailbox IN DEVICE MEMORY (large-BAR systems: the host stores straight into VRAM): every block polls its own
        // HBM -- 0.11 us a poll instead of a 1.1 us PCIe read, and nobody has to pass the challenge on (profiles/r5f_host_visible_vram_probe.txt).
        // The mapping is write-combining: the eight self-validating words are pushed out by the fence.
        volatile uint64_t *vslot = p->d_vram_mail + 8 * (sv & 1u);
        for (int i = 0; i < 8; ++i) vslot[i] = ((uint64_t)(uint32_t)(vm.l[i >> 1] >> (32 * (i & 1))) << 32) | sv;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        return;
    }
    for (int i = 0; i < 8; ++i) {
        const uint32_t limb = (uint32_t)(vm.l[i >> 1] >> (32 * (i & 1)));
        __atomic_store_n(slot + i, ((uint64_t)limb << 32) | sv, __ATOMIC_RELEASE);
    }
}

// n_rounds rounds (prove_round, feed, sample) starting at the handle's next round; r_or_null = the challenge that round binds
int run_tail(sc_prover *p, sch::Blake2b512Rng &rng, uint32_t n_rounds, const sch::Fr *r_or_null, uint64_t *out_msgs, sch::Fr *out_challenges) {
    gate_lock(p->device); // until the kernel is launched; the host loop below makes no HIP calls
    struct Unlock {
        const int device;
        bool held = true;
        void release() {
            if (held) gate_unlock(device);
            held = false;
        }
        ~Unlock() { release(); }
    } gate{p->device};
    HIP_TRY(hipSetDevice(p->device));
    int rc_t = collect_timing(p);
    if (rc_t) return rc_t;
    const uint32_t D = p->D;
    scd::TailArgs A;
    int grid = 1;
    const int slices_B = tail_slices_blocks_for(p);
    int rc_l = tail_launch(p, n_rounds, r_or_null, scd::wait_spins_default(), A, grid, slices_B);
    if (rc_l) return rc_l;
    g_stat[kStatTailLaunches].fetch_add(1, std::memory_order_relaxed);
    if (slices_B > 0) g_stat[kStatTailSlices].fetch_add(1, std::memory_order_relaxed);
    scd::plan_hit(slices_B > 0 ? (p->max_mult > (uint32_t)scd::kMaxFusedM ? scd::kPlanTailSlices12 : scd::kPlanTailSlices8) : scd::kPlanTailRounds);
    gate.release();
    p->seq += n_rounds;
    p->sig_seq += n_rounds - 1;
    if (r_or_null) p->randomness.push_back(*r_or_null); // bound by the first of these rounds (prover.rs:84)
    // the host's half: wait for a message, hash, answer
    static const bool trace = std::getenv("SC_HOST_TRACE") != nullptr; // stderr: arrival time of every tail message
    auto t_prev = std::chrono::steady_clock::now();
    int rc = SC_OK;
    for (uint32_t j = 0; j < n_rounds; ++j) {
        uint64_t *pm = out_msgs + (size_t)j * D * 4;
        if (rc == SC_OK) {
            uint64_t spins = 0;
            bool seen = false;
            const auto t_start = std::chrono::steady_clock::now();
            while (!(seen = (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == A.seq0 + j))) {
                if ((++spins & 0xfff) == 0) {
                    if (wait_gave_up(p)) break;
                    if (std::chrono::steady_clock::now() - t_start > publish_timeout()) break;
                }
            }
            if (!seen || wait_gave_up(p)) {
                if (std::getenv("SC_HOST_TRACE")) {
                    const uint64_t *tg = reinterpret_cast<const uint64_t *>(p->h_mail) + 16;
                    std::fprintf(stderr, "[sc] tail failure: waiting for message %u of %u (seq %u), h_flag %u, give-up marker %u, sig0 %u, grid %d, first_pairs %llu, tags %u %u\n", j,
                                 n_rounds, A.seq0 + j, __atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE), __atomic_load_n(p->sig + 1, __ATOMIC_ACQUIRE), A.sig0, grid,
                                 (unsigned long long)A.first_pairs, (uint32_t)tg[0], (uint32_t)tg[8]);
                }
                rc = sc_internal_fail(SC_ERR_HIP, wait_gave_up(p) ? "the host took longer than the wait kernel's bound to deliver a challenge; the proof is void"
                                                      : "a tail round did not publish its message within 20 s");
            }
        }
        if (trace) {
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[sc] tail round %u (%llu pairs): message after %.1f us\n", p->round + j + 1,
                         (unsigned long long)(A.first_pairs >> j), std::chrono::duration<double, std::micro>(now - t_prev).count());
            t_prev = now;
        }
        sch::Fr vm = sch::zero();
        if (rc == SC_OK) {
            std::memcpy(pm, p->h_out, (size_t)D * 32);
            rng.feed_prover_msg(reinterpret_cast<const sch::Fr *>(pm), D); // mod.rs:61
            vm = rng.sample_fr();                                           // mod.rs:63
            if (out_challenges) out_challenges[j] = vm;
        }
        if (j + 1 < n_rounds) { // (on the error path: a zero challenge, so that the kernel runs to its end and the stream drains)
            tail_post_challenge(p, A.sig0 + j + 1, vm, slices_B > 0 && p->d_vram_mail != nullptr);
            if (rc == SC_OK) p->randomness.push_back(vm);
        }
    }
    if (rc != SC_OK) {
        DeviceGate g2(p->device);
        (void)hipStreamSynchronize(p->stream);
        p->exhausted = true; // tables are no longer meaningful: the handle must be reset
        return rc;
    }
    // the handle's state after the tail: rounds done, challenges bound, where the tables are
    p->round += n_rounds;
    tail_epilogue_tables(p, n_rounds - 1 + (r_or_null ? 1 : 0));
    p->timed = false;
    return SC_OK;
}

// ---- the resident kernel of the INTERACTIVE protocol ---------------------------------------------------------------------------
// IPForMLSumcheck::prove_round called round by round (prover.rs:74-77; mod.rs:59-64 with a caller's own FeedableRNG) pays a launch
// sequence per late round: bind, sums, finalize -- 26-30 us for a few microseconds of arithmetic.  Instead, the first late-round call
// launches the persistent tail kernel for ALL remaining rounds and returns its first message; the kernel stays on the GPU polling the
// host-mapped mailbox, and every following sc_prove_round only posts its challenge and waits for the next message.  The kernel's
// patience is short (resident_spins polls, ~0.5 ms): a verifier that does not answer in time finds the kernel gone -- it leaves cleanly
// after the last round it completed, tables consistent -- and the call proceeds as if there had never been one (a launch sequence, or a
// new resident kernel).  Every other entry point that touches the handle's stream or tables quiesces it first (a tagged stop word).
void resident_release_slot(sc_prover *p) { tail_slot_release(p); }
// the kernel has exited (all rounds done, patience expired, or stop word): fold what it did into the handle
int resident_finish(sc_prover *p) {
    if (!p->res.active) return SC_OK;
    DeviceGate gate_(p->device);
    (void)hipSetDevice(p->device);
    const hipError_t e = hipStreamSynchronize(p->stream);
    const sc_prover::Resident r = p->res;
    p->res.active = false;
    p->seq = r.seq0 - 1 + r.done;
    p->sig_seq = r.sig0 + r.done; // past every tag a word of the mailbox may carry (an unconsumed challenge, the stop word)
    if (p->sig) __atomic_store_n(p->sig + 1, 0u, __ATOMIC_RELEASE); // its exit marker is not a voided proof
    resident_release_slot(p);
    if (e != hipSuccess) {
        p->exhausted = true;
        return sc_internal_fail(SC_ERR_HIP, "the resident round kernel failed: %s", hipGetErrorString(e));
    }
    if (r.done == 0) { // it never published: the tables may be half bound
        p->exhausted = true;
        return sc_internal_fail(SC_ERR_HIP, "the resident round kernel left before its first message");
    }
    tail_epilogue_tables(p, r.done - 1 + (r.first_has_bind ? 1 : 0));
    p->timed = false;
    return SC_OK;
}
// ask it to leave (any entry point other than sc_prove_round), then fold
int resident_quiesce(sc_prover *p) {
    if (!p || !p->res.active) return SC_OK;
    if (p->res.done < p->res.n_rounds) { // it is (or will be) polling for the challenge tagged sig0 + done: word 0 with the stop bit
        const uint32_t want = p->res.sig0 + p->res.done;
        uint64_t *slot = reinterpret_cast<uint64_t *>(p->h_mail) + 16 + 8 * (want & 1u);
        __atomic_store_n(slot, (uint64_t)(want ^ 0x80000000u), __ATOMIC_RELEASE);
    }
    return resident_finish(p);
}
// wait for the message of the kernel's round j.  SC_OK: in out_evals; kResidentGone: the kernel left before computing it
int resident_wait(sc_prover *p, uint32_t j, uint64_t *out_evals) {
    const uint32_t want = p->res.seq0 + j;
    uint64_t spins = 0;
    const auto t_start = std::chrono::steady_clock::now();
    for (;;) {
        if (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == want) break;
        if ((++spins & 0xff) == 0) {
            if (wait_gave_up(p)) { // (re-check the flag: the message may have been published just before an exit for another reason)
                if (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == want) break;
                int rc = resident_finish(p);
                g_stat[kStatResidentGone].fetch_add(1, std::memory_order_relaxed);
                return rc ? rc : kResidentGone;
            }
            if (std::chrono::steady_clock::now() - t_start > publish_timeout()) {
                (void)resident_quiesce(p);
                p->exhausted = true;
                return sc_internal_fail(SC_ERR_HIP, "the resident round kernel did not publish its message within 20 s");
            }
        }
    }
    std::memcpy(out_evals, p->h_out, (size_t)p->D * 32);
    p->res.done = j + 1;
    p->round += 1;
    if (p->res.done == p->res.n_rounds) return resident_finish(p); // the last round: the kernel ends by itself
    return SC_OK;
}
bool resident_enabled(sc_prover *p) {
    return scd::policy(scd::kPolResident) != 0 && p->resident_spins > 0 && !p->timing && p->stream == p->own_stream;
}
// first late-round call: launch the kernel for every remaining round, return its first message
int resident_start(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals) {
    // argument errors keep the reference's precedence: the ordinary path reports them
    if ((r_or_null && p->round == 0) || (!r_or_null && p->round > 0)) return kResidentGone;
    sch::Fr r = sch::zero();
    if (r_or_null) {
        std::memcpy(&r, r_or_null, 32);
        if (sch::geq_p(r)) return kResidentGone;
    }
    if (!resident_enabled(p) || !tail_possible(p, true)) return kResidentGone;
    if (!tail_slot_acquire(p, true)) return kResidentGone;
    const uint32_t n_rounds = p->nv - p->round;
    scd::TailArgs A;
    int grid = 1;
    {
        DeviceGate gate_(p->device);
        // (k_tail_slices where the shape allows, its block 0 the only one that listens to the host: one poller decides for all whether a
        // challenge came in time, and on the way out every block writes its slice back)
        int rc = SC_ERR_HIP, slices_B = 0;
        if (hipSetDevice(p->device) == hipSuccess) {
            slices_B = tail_slices_blocks_for(p);
            rc = tail_launch(p, n_rounds, r_or_null ? &r : nullptr, p->resident_spins, A, grid, slices_B, true);
        }
        if (rc == SC_OK && slices_B > 0) g_stat[kStatTailSlices].fetch_add(1, std::memory_order_relaxed);
        if (rc == SC_OK) scd::plan_hit(slices_B > 0 ? scd::kPlanResidentSlices : scd::kPlanResidentRounds);
        if (rc) {
            resident_release_slot(p);
            return rc;
        }
    }
    g_stat[kStatResidentStarts].fetch_add(1, std::memory_order_relaxed);
    p->res.active = true;
    p->res.first_has_bind = r_or_null != nullptr;
    p->res.seq0 = A.seq0;
    p->res.sig0 = A.sig0;
    p->res.n_rounds = n_rounds;
    p->res.done = 0;
    if (r_or_null) p->randomness.push_back(r);
    int rc = resident_wait(p, 0, out_evals);
    if (rc == kResidentGone) { // cannot be: round 0 of the kernel waits for nobody
        p->exhausted = true;
        return sc_internal_fail(SC_ERR_HIP, "the resident round kernel left before its first message");
    }
    return rc;
}
// a following call: post the challenge, take the next message
int resident_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals) {
    if (!r_or_null) return sc_internal_fail(SC_ERR_MISSING_MSG, "verifier message is empty"); // (round > 0 here; the kernel keeps waiting)
    sch::Fr r;
    std::memcpy(&r, r_or_null, 32);
    if (sch::geq_p(r)) return sc_internal_fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
    const uint32_t j = p->res.done; // the kernel's round this call completes
    tail_post_challenge(p, p->res.sig0 + j, r);
    const size_t n_rand = p->randomness.size();
    p->randomness.push_back(r);
    int rc = resident_wait(p, j, out_evals);
    if (rc == kResidentGone) p->randomness.resize(n_rand); // the ordinary path records it again
    return rc;
}

// Rounds first..last-1 (0-based) of the reference's prove loop (mod.rs:57-64): prove_round, feed, sample.  Late rounds are
// pipelined: while round i runs, round i+1 is already enqueued behind the wait, so hashing round i's message and storing the
// challenge is all that separates the two on the critical path.  vm/have carry the pending challenge in and out.
int run_rounds(sc_prover *p, sch::Blake2b512Rng &rng, uint32_t n_rounds, uint64_t *out_msgs, sch::Fr *out_challenges_or_null,
                      double *t_launch, double *t_wait, double *t_fs) {
    using clk = std::chrono::steady_clock;
    const uint32_t D = p->D;
    sch::Fr vm = sch::zero();
    bool have = false, enqueued = false;
    uint32_t want = 0;
    for (uint32_t i = 0; i < n_rounds; ++i) {
        uint64_t *pm = out_msgs + (size_t)i * D * 4;
        const auto t0 = clk::now();
        int rc;
        if (!enqueued && tail_possible(p, true)) { // from here on every round is latency-bound: one persistent kernel runs them all
            TailSlot slot(p);                // (unless another prover's tail kernel has the device: then pipelined launches, below)
            if (slot.held) return run_tail(p, rng, n_rounds - i, have ? &vm : nullptr, pm, out_challenges_or_null ? out_challenges_or_null + i : nullptr);
        }
        if (!enqueued) {
            rc = launch_round(p, have ? vm.l : nullptr, nullptr, true);
            if (rc) return rc;
            want = p->seq;
        }
        uint32_t want_next = 0;
        bool next_enqueued = false;
        // round i+1 goes in now, behind the wait -- unless it is one the persistent tail kernel will take (it starts after round i's challenge)
        const bool next_is_tail = tail_shape_ok(p) && p->round < p->nv && !(p->streamed && p->round < 2) &&
                                  ((1ULL << (p->nv - (p->round + 1))) <= scd::kTailMaxPairs ||
                                   ((1ULL << (p->nv - (p->round + 1))) <= scd::kTsMaxPairs && tail_slices_blocks_for(p) > 0));
        if (i + 1 < n_rounds && !next_is_tail && can_defer_next(p)) {
            gate_lock(p->device); // held until the challenge is handed over: see DeviceGate
            rc = launch_round(p, nullptr, nullptr, true, true);
            if (rc) {
                gate_unlock(p->device);
                return rc;
            }
            want_next = p->seq;
            next_enqueued = true;
        }
        const auto t1 = clk::now();
        rc = await_round(p, pm, want);
        if (rc) {
            if (next_enqueued) gate_unlock(p->device);
            return rc;
        }
        const auto t2 = clk::now();
        rng.feed_prover_msg(reinterpret_cast<const sch::Fr *>(pm), D); // mod.rs:61
        vm = rng.sample_fr();                                           // mod.rs:63
        have = true;
        if (out_challenges_or_null) out_challenges_or_null[i] = vm;
        if (next_enqueued) {
            provide_challenge(p, vm);
            gate_unlock(p->device);
        }
        enqueued = next_enqueued;
        want = want_next;
        if (t_launch) {
            const auto t3 = clk::now();
            *t_launch += std::chrono::duration<double, std::micro>(t1 - t0).count();
            *t_wait += std::chrono::duration<double, std::micro>(t2 - t1).count();
            *t_fs += std::chrono::duration<double, std::micro>(t3 - t2).count();
        }
    }
    return SC_OK;
}

extern "C" int sc_prove_round_partial(sc_prover *p, const uint64_t *r_or_null, uint64_t *d_wide_out) {
    if (!p || !d_wide_out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    return launch_round(p, r_or_null, d_wide_out, false);
}

// ---------------------------------------------------------------------------------------------------
// MLSumcheck::prove_as_subprotocol (reference src/ml_sumcheck/mod.rs:50-70)
// ---------------------------------------------------------------------------------------------------
// The Fiat-Shamir loop of mod.rs:54-67 on an existing handle at round 0 (fresh from sc_prover_init or sc_prover_reset).
// gkr.hip: one sumcheck phase's rounds through the same (pipelined) loop
int sc_internal_run_rounds(sc_prover *p, sch::Blake2b512Rng &rng, uint32_t n_rounds, uint64_t *out_msgs, sch::Fr *out_challenges) {
    double a = 0, b = 0, c = 0;
    int rc = run_rounds(p, rng, n_rounds, out_msgs, out_challenges, nullptr, &b, &c);
    (void)a;
    if (rc) abandon_deferred(p);
    return rc;
}

// n_rounds of the prove loop on a handle at round 0, continuing `rng` (no PolynomialInfo is fed): the tail of a sharded proof
extern "C" int sc_ml_prove_rounds(sc_prover *p, sc_rng *rng, uint32_t n_rounds, uint64_t *out_proof, uint64_t *out_randomness) {
    if (!p || !rng || !out_proof || !out_randomness) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (p->round != 0 || n_rounds > p->nv) return sc_internal_fail(SC_ERR_BAD_ARG, "handle must be at round 0 and hold at least n_rounds variables");
    std::vector<sch::Fr> ch(n_rounds);
    int rc = sc_internal_run_rounds(p, rng->rng, n_rounds, out_proof, ch.data());
    if (rc) return rc;
    if (n_rounds) std::memcpy(out_randomness, ch.data(), (size_t)n_rounds * 32);
    return SC_OK;
}

extern "C" int sc_ml_prove_handle(sc_prover *p, sc_rng *rng_or_null, uint64_t *out_proof) {
    if (!p || !out_proof) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (p->round != 0) return sc_internal_fail(SC_ERR_BAD_ARG, "handle is not at round 0");
    sc_rng local;
    sch::Blake2b512Rng &rng = rng_or_null ? rng_or_null->rng : local.rng;
    rng.feed_poly_info(p->max_mult, p->nv); // mod.rs:54
    static const bool trace = std::getenv("SC_HOST_TRACE") != nullptr; // stderr: where the host's share of a proof goes
    double t_launch = 0, t_wait = 0, t_fs = 0;
    std::vector<sch::Fr> ch(p->nv);
    const sch::Blake2b512Rng transcript_at_start = rng;
    int rc = run_rounds(p, rng, p->nv, out_proof, ch.data(), trace ? &t_launch : nullptr, &t_wait, &t_fs);
    if (rc && wait_gave_up(p) && (p->borrow || p->streamed || p->host_tabs.size() == p->U)) {
        // A device-side wait expired (something stalled this thread or its HIP calls for longer than the bound): the rounds after
        // it ran on a stale challenge.  The inputs are intact, so prove again from round 0 with every round synchronous.
        abandon_deferred(p);
        const bool was = p->pipeline_ok;
        if (sc_prover_reset(p, nullptr, 0) == SC_OK) {
            if (trace) std::fprintf(stderr, "[sc] a device-side wait expired; proving again without pipelining\n");
            p->pipeline_ok = false;
            rng = transcript_at_start;
            rc = run_rounds(p, rng, p->nv, out_proof, ch.data(), trace ? &t_launch : nullptr, &t_wait, &t_fs);
            p->pipeline_ok = was;
            ++p->n_retries;
            g_stat[kStatProofRetries].fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (rc) {
        abandon_deferred(p);
        return rc;
    }
    const sch::Fr vm = p->nv ? ch[p->nv - 1] : sch::zero();
    if (trace) std::fprintf(stderr, "[sc] proof host time: launch %.1f us, wait %.1f us, transcript %.1f us (%u rounds)\n", t_launch, t_wait, t_fs, p->nv);
    p->randomness.push_back(vm); // mod.rs:65-67: recorded, never bound
    return SC_OK;
}

extern "C" int sc_ml_prove(const sc_poly_desc *desc, sc_rng *rng_or_null, uint64_t *out_proof, sc_prover **out_state_or_null) {
    if (!desc || !out_proof) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (out_state_or_null) *out_state_or_null = nullptr;
    int rc = validate_desc(desc); // prover_init panics on a constant before anything is proved (prover.rs:50-52)
    if (rc) return rc;
    sc_poly_desc eff = *desc;
    // Without a state to hand back the prover does not outlive this call, and it never writes a caller's table: device tables are
    // read in place instead of being copied first (their producers are waited for, as a copy would).
    if (!out_state_or_null && (eff.flags & SC_TABLES_ON_DEVICE) && !(eff.flags & SC_TABLES_BORROW)) {
        if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
        const int dev_ = sc_internal_device_ref();
        DeviceGate gate_(dev_);
        HIP_TRY(hipSetDevice(dev_));
        HIP_TRY(hipDeviceSynchronize());
        eff.flags |= SC_TABLES_BORROW;
    }
    sc_prover *p = nullptr;
    rc = sc_prover_init(&eff, &p); // (takes the kept prover when the structure matches)
    if (rc) return rc;
    rc = sc_ml_prove_handle(p, rng_or_null, out_proof);
    if (rc) {
        p->pool_key.clear();
        prover_destroy(p);
        return rc;
    }
    if (out_state_or_null) *out_state_or_null = p;
    else sc_prover_free(p);
    return SC_OK;
}
// The first n_rounds rounds of MLSumcheck::prove_as_subprotocol (reference src/ml_sumcheck/mod.rs:54-64) on this rank's shard:
// per round the shard's kernels, one all-reduce (sum, uint64) of the (deg+1) x 8 zero-extended limbs -- ncclAllReduce on the same
// stream, or the host transport's function on the published lanes -- then, on every rank identically, fold, feed, sample.  The
// handle is left after round n_rounds (its tables have two entries when n_rounds == its num_vars).
int sharded_rounds(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, uint32_t n_rounds, uint64_t *out_proof, uint64_t *out_randomness) {
    HIP_TRY(hipSetDevice(p->device));
    const int n_words = (int)p->D * 8;
    // (before anything changes the handle: the peer-to-peer inbox holds kP2PWords lanes per source, i.e. messages of at most 8 evaluations)
    if (comm->p2p && comm->nranks > 1 && n_words > scd::kP2PWords)
        return sc_internal_fail(SC_ERR_BAD_ARG, "a peer-to-peer communicator carries round messages of at most %d evaluations (max_multiplicands <= %d); this polynomial has %u",
                    scd::kP2PWords / 8, scd::kP2PWords / 8 - 1, p->D);
    if (!p->d_wide) {
        HIP_TRY(hipMalloc(&p->d_wide, (size_t)n_words * 8));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_wide), (size_t)n_words * 8, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_wide_dev), p->h_wide, 0));
        std::memset(p->h_wide, 0, (size_t)n_words * 8); // (no stale word may look like a tagged one)
    }
    const bool p2p = comm->p2p != nullptr && comm->nranks > 1;
    const bool on_stream = comm->comm != nullptr || p2p; // RCCL / p2p: the reduction is a stream operation between the round and its publication
    // RCCL, direct publication (comm.hip: rccl_direct_probe): the finalize step tags its lanes with the round's sequence number, the
    // all-reduce writes its result straight into the host-mapped page, and a word whose top bits read nranks * tag IS this round's
    // total -- the per-round sequence on the stream is [round kernel, finalize, ncclAllReduce] with no publish launch behind it (the
    // peer-to-peer communicator's exchange kernel publishes too: one launch of latency per sharded round on either path).
    // The decision is the COMMUNICATOR's (agreed by every rank inside sc_comm_init), never this rank's alone: a rank that tagged while a
    // peer did not would wait for a total that cannot come.  Streamed rounds (whose message k_msg_accumulate forms, untagged) get their tag
    // from one more small launch.
    const bool direct = comm->comm != nullptr && comm->direct_publish;
    if (direct) std::memset(p->h_wide, 0, (size_t)n_words * 8); // (nothing targets the page now; a stale word of an earlier communicator or generation cycle must not look ready)
    struct TagScope {
        sc_prover *p;
        ~TagScope() { p->wide_tagged = false; }
    } tag_scope_{p};
    p->wide_tagged = direct;
    scd::plan_hit(direct ? scd::kPlanShardedRcclDirect : comm->comm ? scd::kPlanShardedRcclPublish : p2p ? scd::kPlanShardedP2P : scd::kPlanShardedHost);
    const uint64_t lane_mask = (1ULL << scd::kWideTagShift) - 1;
    uint32_t gen_want = 0; // the communicator's generation of the round awaited (ranks make the same calls: the same on every rank)
    auto wide_ready = [&](uint32_t gen) -> bool { // every word carries nranks * the tag every rank's finalize step gave that generation
        const uint64_t expect = (uint64_t)comm->nranks * scd::wide_tag_of(gen);
        for (int w = n_words - 1; w >= 0; --w)
            if ((__atomic_load_n(p->h_wide + w, __ATOMIC_ACQUIRE) >> scd::kWideTagShift) != expect) return false;
        return true;
    };
    // Pipelined late rounds park a polling kernel on the stream until THIS rank's host has the next challenge -- which needs every
    // rank's lanes.  RCCL ranks sit on distinct devices.  Host-transport ranks may share one GPU (tests; threads of one process), where
    // streams share hardware queues: rank A's polling kernel could then sit in front of rank B's round kernels, and A's host would
    // wait for B forever.  So the rounds are only pipelined where no other rank's work can queue behind the wait (the same goes for a
    // p2p group with two ranks on one GPU, whose exchange kernel also gives up quickly and is launched again by the host loop below).
    const bool may_defer = comm->comm != nullptr || comm->nranks == 1 || (p2p && !comm->p2p_shared_device);
    sch::Fr vm = sch::zero();
    bool have = false, enqueued = false;
    uint32_t want = 0;
    std::vector<uint64_t> evals((size_t)p->D * 4);
    scd::P2PArgs xargs[2]; // the exchange of the round awaited (slot want & 1) and of the pipelined one behind it
    auto fill_xargs = [&](scd::P2PArgs &a) {
        std::memset(&a, 0, sizeof(a));
        for (int q = 0; q < comm->nranks; ++q) a.inbox[q] = comm->p2p->inbox[q];
        a.nranks = comm->nranks;
        a.rank = comm->rank;
        a.n_words = n_words;
        a.gen = ++comm->p2p_gen;
        a.max_spins = comm->p2p_shared_device ? 2048u : scd::wait_spins_default();
    };
    // one round on the stream: local kernels -> d_wide, integer all-reduce in place, publish to the host-mapped page.  With
    // deferred = true the whole sequence sits behind the wait kernel (pipelined late rounds, see run_rounds): every rank's host
    // derives the same challenge at about the same time, so the ranks' all-reduces still meet.
    auto enqueue = [&](const uint64_t *r, bool deferred, uint32_t *want_out, uint32_t *gen_out) -> int {
        DeviceGate gate_(p->device);
        *gen_out = p->wide_gen = direct ? ++comm->direct_gen : 0;
        const bool streamed_round = p->streamed && p->round < 2;
        int rc = launch_round(p, r, p->d_wide, false, deferred);
        if (rc) return rc;
        if (direct && streamed_round) HIP_TRY(scd::launch_tag_words(p->d_wide, n_words, *gen_out, p->stream));
        if (comm->comm) NCCL_TRY(g_nccl.AllReduce(p->d_wide, direct ? p->h_wide_dev : p->d_wide, (size_t)n_words, ncclUint64, ncclSum, comm->comm, p->stream));
        p->seq += 1;
        *want_out = p->seq;
        if (p2p) { // the all-reduce and the publication are one kernel
            fill_xargs(xargs[*want_out & 1u]);
            HIP_TRY(scd::launch_p2p_allreduce(xargs[*want_out & 1u], p->d_wide, p->h_wide_dev, p->h_flag_dev, *want_out, p->stream));
        } else if (!direct) {
            HIP_TRY(scd::launch_publish_words(p->d_wide, p->h_wide_dev, n_words, p->h_flag_dev, *want_out, p->stream));
        }
        return SC_OK;
    };
    for (uint32_t i = 0; i < n_rounds; ++i) {
        int rc;
        if (!enqueued && (rc = enqueue(have ? vm.l : nullptr, false, &want, &gen_want))) return rc;
        uint32_t want_next = 0, gen_next = 0;
        bool next_enqueued = false;
        if (i + 1 < n_rounds && may_defer && can_defer_next(p)) {
            gate_lock(p->device); // held until the challenge is handed over: see DeviceGate
            if ((rc = enqueue(nullptr, true, &want_next, &gen_next))) {
                abandon_deferred(p);
                gate_unlock(p->device);
                return rc;
            }
            next_enqueued = true;
        }
        struct GateRelease { // every early return below leaves the window
            const int device;
            bool held;
            ~GateRelease() {
                if (held) gate_unlock(device);
            }
        } window{p->device, next_enqueued};
        uint64_t spins = 0;
        bool seen = false;
        const auto t_start = std::chrono::steady_clock::now();
        while (!(seen = direct ? wide_ready(gen_want) : (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == want))) {
            if (p2p && __atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == (want | scd::kP2PRetryBit)) {
                // the exchange kernel left without its peers' words.  Ranks that share a GPU: a peer's kernels may have been queued behind
                // it -- launch it again (pushes are idempotent, what has arrived stays).  Distinct GPUs: its bound is seconds; a peer is gone.
                if (!comm->p2p_shared_device || next_enqueued || std::chrono::steady_clock::now() - t_start > publish_timeout()) break;
                __atomic_store_n(p->h_flag, 0u, __ATOMIC_RELEASE);
                std::this_thread::yield();
                DeviceGate gate_(p->device);
                HIP_TRY(scd::launch_p2p_allreduce(xargs[want & 1u], p->d_wide, p->h_wide_dev, p->h_flag_dev, want, p->stream));
                continue;
            }
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t_start > publish_timeout()) break;
        }
        if (!seen && p2p && (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) & scd::kP2PRetryBit)) {
            abandon_deferred(p);
            p->exhausted = true;
            return sc_internal_fail(SC_ERR_HIP, "p2p all-reduce: a peer's lanes did not arrive");
        }
        if (!seen) {
            if (p->deferred_pending) {
                abandon_deferred(p);
                return sc_internal_fail(SC_ERR_HIP, "sharded round did not publish within 20 s");
            }
            HIP_TRY(hipStreamSynchronize(p->stream));
            if (!(direct ? wide_ready(gen_want) : (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == want)))
                return sc_internal_fail(SC_ERR_HIP, "sharded round finished without publishing");
        }
        if (wait_gave_up(p)) {
            abandon_deferred(p);
            p->exhausted = true;
            return sc_internal_fail(SC_ERR_HIP, "the host took longer than the wait kernel's bound to deliver a challenge; the proof is void");
        }
        std::vector<uint64_t> lanes(p->h_wide, p->h_wide + n_words); // (the device reuses the page for the next round)
        if (direct)
            for (uint64_t &w : lanes) w &= lane_mask;
        if (!on_stream && comm->nranks > 1) {
            GateYield yield_(p->device, true);
            if (comm->h_allreduce(comm->ctx, lanes.data(), (size_t)n_words) != 0) {
                abandon_deferred(p);
                return sc_internal_fail(SC_ERR_HIP, "the host transport's all-reduce failed");
            }
        }
        rc = sc_wide_reduce(lanes.data(), p->D, evals.data());
        if (rc) {
            abandon_deferred(p);
            return rc;
        }
        std::memcpy(out_proof + (size_t)i * p->D * 4, evals.data(), (size_t)p->D * 32);
        rng.feed_prover_msg(reinterpret_cast<const sch::Fr *>(evals.data()), p->D); // mod.rs:61
        vm = rng.sample_fr();                                                       // mod.rs:63
        have = true;
        std::memcpy(out_randomness + (size_t)i * 4, vm.l, 32);
        if (next_enqueued) provide_challenge(p, vm);
        enqueued = next_enqueued;
        want = want_next;
        gen_want = gen_next;
        // (`window` releases the gate here)
    }
    return SC_OK;
}

extern "C" int sc_ml_prove_sharded_rounds(sc_prover *p, sc_comm *comm, sc_rng *rng, uint32_t nv_total, uint32_t n_rounds, uint64_t *out_proof,
                                          uint64_t *out_randomness) {
    if (!p || !comm || !rng || !out_proof || !out_randomness) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (p->round != 0 || n_rounds > p->nv) return sc_internal_fail(SC_ERR_BAD_ARG, "handle must be at round 0 and hold at least n_rounds variables");
    rng->rng.feed_poly_info(p->max_mult, nv_total); // mod.rs:54: the GLOBAL instance's info
    return sharded_rounds(p, comm, rng->rng, n_rounds, out_proof, out_randomness);
}

// The tail of a sharded proof.  Sharded rounds pay a collective each; once the GLOBAL instance is down to a latency-bound size
// (at most 2^14 pairs) it is cheaper to stop sharding: every rank binds the last challenge into what is left of its shard (2^m
// entries per table), the remainders are all-gathered (U * 2^m * 32 bytes per rank), and every rank finishes the last m + log2 G
// rounds on the same complete tables -- no exchange any more (the transcripts stay in step: they absorb identical messages), at the
// single-GPU cost per round, in the persistent tail kernel.  m = 0 (one element per table and rank) is the smallest case.
uint32_t sharded_tail_m(uint32_t nv_local, uint32_t k) { // log2 of the entries per table a rank still holds at the gather
    if (k == 0) return 0;
    // (sc_set_policy("shard_gather_log2", v), default 15 = 2^14 pairs in the first replicated round: the same on every rank.  A smaller value trades
    // gather volume -- U * 2^value * 32 bytes in all -- for sharded rounds, each an exchange; to be tuned on the first multi-GPU node)
    const uint32_t glog = (uint32_t)std::min<int64_t>(std::max<int64_t>(scd::policy(scd::kPolShardGatherLog2), 1), 15);
    const uint32_t want = k >= glog ? 0u : glog - k; // 2^(m + k - 1) pairs <= 2^(glog - 1) in the first replicated round
    return std::min(want, nv_local - 1);          // at least one sharded round
}
int sharded_tail(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, const uint64_t *last_challenge, uint32_t k, uint32_t m, uint64_t *out_proof,
                        uint64_t *out_randomness) {
    scd::plan_hit(scd::kPlanShardedGatherTail);
    const uint32_t G = (uint32_t)comm->nranks, U = p->U, per = 1u << m;
    const size_t send_bytes = (size_t)U * per * 32;
    int rc;
    {
        DeviceGate gate_(p->device); // (not across the replicated rounds below: their calls take it themselves, and a persistent tail
                                     // kernel's host loop must never hold it)
        HIP_TRY(hipSetDevice(p->device));
        if (p->tail && (p->tail->nv != k + m || p->tail_ranks != G)) { // the handle meets a communicator of another size: rebuild the tail
            prover_destroy(p->tail);
            p->tail = nullptr;
        }
        // the three exchange buffers follow the sizes of THIS call, whatever an earlier (possibly failed) call left behind
        if (p->tail_buf_bytes != send_bytes * G || p->tail_ranks != G) {
            if (p->tail) { // it borrows d_tail_tabs
                prover_destroy(p->tail);
                p->tail = nullptr;
            }
            (void)hipFree(p->d_tail_send);
            (void)hipFree(p->d_tail_recv);
            (void)hipFree(p->d_tail_tabs);
            p->d_tail_send = p->d_tail_recv = p->d_tail_tabs = nullptr;
            p->tail_buf_bytes = 0;
            HIP_TRY(hipMalloc(&p->d_tail_send, send_bytes));
            HIP_TRY(hipMalloc(&p->d_tail_recv, send_bytes * G));
            HIP_TRY(hipMalloc(&p->d_tail_tabs, send_bytes * G));
            p->tail_buf_bytes = send_bytes * G;
        }
        p->tail_ranks = G;
        rc = prover_bind_out(p, last_challenge, reinterpret_cast<uint64_t *>(p->d_tail_send));
        if (rc) return rc;
        if (comm->comm) {
            NCCL_TRY(g_nccl.AllGather(p->d_tail_send, p->d_tail_recv, send_bytes / 8, ncclUint64, comm->comm, p->stream));
        } else if (comm->p2p && comm->nranks > 1) {
            rc = p2p_allgather(comm, p->d_tail_send, p->d_tail_recv, send_bytes, p->stream);
            if (rc) return rc;
        } else {
            std::vector<uint64_t> send(send_bytes / 8), recv(send_bytes / 8 * G);
            HIP_TRY(hipMemcpyAsync(send.data(), p->d_tail_send, send_bytes, hipMemcpyDeviceToHost, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream));
            {
                GateYield yield_(p->device, comm->nranks > 1);
                if (comm->h_allgather(comm->ctx, send.data(), recv.data(), send_bytes) != 0) return sc_internal_fail(SC_ERR_HIP, "the host transport's all-gather failed");
            }
            HIP_TRY(hipMemcpyAsync(p->d_tail_recv, recv.data(), send_bytes * G, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream)); // `recv` goes out of scope
        }
        HIP_TRY(scd::launch_gather_to_tables(static_cast<const uint4 *>(p->d_tail_recv), static_cast<uint4 *>(p->d_tail_tabs), G, U, per, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream)); // the tail prover runs on its own stream
        if (!p->tail) { // built once per handle: the same products over U borrowed tables of G * 2^m entries
            std::vector<uint64_t> coeffs((size_t)p->K * 4);
            std::vector<uint32_t> offs(1, 0), idx;
            for (uint32_t q = 0; q < p->K; ++q) {
                std::memcpy(&coeffs[4 * q], &p->prods[q].coeff, 32);
                idx.insert(idx.end(), p->prod_indices[q].begin(), p->prod_indices[q].end());
                offs.push_back((uint32_t)idx.size());
            }
            std::vector<const uint64_t *> tabs(U);
            for (uint32_t u = 0; u < U; ++u) tabs[u] = reinterpret_cast<const uint64_t *>(static_cast<char *>(p->d_tail_tabs) + (size_t)u * G * per * 32);
            sc_poly_desc d;
            std::memset(&d, 0, sizeof(d));
            d.num_vars = k + m;
            d.max_multiplicands = p->max_mult;
            d.n_products = p->K;
            d.coeffs = coeffs.data();
            d.prod_offsets = offs.data();
            d.prod_indices = idx.data();
            d.n_tables = U;
            d.tables = tabs.data();
            d.flags = SC_TABLES_ON_DEVICE | SC_TABLES_BORROW;
            int &g_device = sc_internal_device_ref();
            const int saved = g_device;
            g_device = p->device;
            rc = sc_prover_init(&d, &p->tail);
            g_device = saved;
            if (rc) return rc;
        } else {
            rc = sc_prover_reset(p->tail, nullptr, 0);
            if (rc) return rc;
        }
    }
    // (same reasoning as in sharded_rounds: no polling kernels on a GPU that other ranks of a host transport may share)
    p->tail->pipeline_ok = !p->polling_off_by_caller; // (the replicated rounds follow the shard handle's sc_prover_set_polling)
    if (!comm->comm && comm->nranks > 1 && !(comm->p2p && !comm->p2p_shared_device)) p->tail->pipeline_ok = false;
    std::vector<sch::Fr> ch(k + m);
    rc = sc_internal_run_rounds(p->tail, rng, k + m, out_proof, ch.data());
    if (rc) return rc;
    std::memcpy(out_randomness, ch.data(), (size_t)(k + m) * 32);
    return SC_OK;
}

// everything of sc_ml_prove_sharded after the transcript has absorbed what precedes the rounds
int sharded_proof_body(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, uint32_t nv_total, uint64_t *out_proof, uint64_t *out_randomness) {
    const uint32_t G = (uint32_t)comm->nranks;
    if (G == 0 || (G & (G - 1)) != 0) return sc_internal_fail(SC_ERR_BAD_ARG, "the number of ranks must be a power of two");
    uint32_t k = 0;
    while ((1u << k) < G) ++k;
    if (p->round != 0 || p->nv + k != nv_total) return sc_internal_fail(SC_ERR_BAD_ARG, "handle must be at round 0 and hold a 1/%u shard of %u variables", G, nv_total);
    uint32_t m = sharded_tail_m(p->nv, k);
    // a streamed shard's tables only exist in HBM once round 2 has bound them: at least two local rounds before the gather (every rank
    // of a group uses the same kind of handle, so every rank computes the same m)
    if (p->streamed && p->nv >= 2) m = std::min(m, p->nv - 2);
    const uint32_t nl = p->nv - m; // nl sharded rounds, then m + k replicated ones
    int rc = sharded_rounds(p, comm, rng, nl, out_proof, out_randomness);
    if (rc) return rc;
    const uint64_t *last = out_randomness + (size_t)(nl - 1) * 4;
    if (k == 0) return sc_prover_push_randomness(p, last); // mod.rs:65-67
    return sharded_tail(p, comm, rng, last, k, m, out_proof + (size_t)nl * p->D * 4, out_randomness + (size_t)nl * 4);
}
extern "C" int sc_ml_prove_sharded(sc_prover *p, sc_comm *comm, sc_rng *rng_or_null, uint32_t nv_total, uint64_t *out_proof, uint64_t *out_randomness) {
    if (!p || !comm || !out_proof || !out_randomness) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    sc_rng local;
    sch::Blake2b512Rng &rng = rng_or_null ? rng_or_null->rng : local.rng;
    rng.feed_poly_info(p->max_mult, nv_total); // mod.rs:54: the GLOBAL instance's info
    return sharded_proof_body(p, comm, rng, nv_total, out_proof, out_randomness);
}
// gkr.hip: the rounds of one GKR sumcheck phase over a sharded pair of tables -- GKRRoundSumcheck::prove feeds no PolynomialInfo
// (gkr_round_sumcheck/mod.rs:108-133) -- and the communicator's shape
int sc_internal_sharded_phase(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, uint32_t nv_total, uint64_t *out_proof, uint64_t *out_randomness) {
    if (!p || !comm || !out_proof || !out_randomness) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    return sharded_proof_body(p, comm, rng, nv_total, out_proof, out_randomness);
}
int sc_internal_comm_rank(sc_comm *c) { return c ? c->rank : 0; }
