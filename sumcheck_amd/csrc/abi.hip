// abi.hip -- the C ABI of libsumcheck_hip.so (declared in include/sumcheck_hip.h), part 1: error plumbing, the device gate, the
// transcript handle, prover handles resident in HBM (build, free, state, reset, the handle pool) and the stand-alone operations
// (fix_variables, evaluate, the verifier, lane folding).  The round launch plan and the protocol loops are in protocol.hip, the
// communicators in comm.hip.
#include "prover_internal.hpp"

// ---------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------
// How long a host loop waits for a round's message (or a peer's lanes) before it declares the proof dead: sc_set_publish_timeout_ms,
// SC_PUBLISH_TIMEOUT_MS in the environment, 20 s by default (a device-side wait's own bound -- policy "wait_spins" -- expires long before).
static std::atomic<uint32_t> g_publish_timeout_ms{0}; // 0: not set yet
std::chrono::milliseconds publish_timeout() {
    uint32_t ms = g_publish_timeout_ms.load(std::memory_order_relaxed);
    if (ms == 0) {
        const char *e = std::getenv("SC_PUBLISH_TIMEOUT_MS");
        const long v = e ? std::atol(e) : 0;
        ms = v > 0 ? (uint32_t)std::min<long>(v, 3600 * 1000L) : 20000u;
        g_publish_timeout_ms.store(ms, std::memory_order_relaxed);
    }
    return std::chrono::milliseconds(ms);
}
extern "C" int sc_set_publish_timeout_ms(uint32_t ms) {
    g_publish_timeout_ms.store(ms ? ms : 20000u, std::memory_order_relaxed);
    return SC_OK;
}
static thread_local std::string g_last_error;
static thread_local int g_device = 0;

// The device gate.  Pipelined rounds leave a kernel in the stream that waits for THIS thread's answer; while it waits, a HIP call of
// this thread must not block.  Measured: with a second thread of the process making HIP calls on the same device, a kernel launch
// behind the waiting kernel did block -- until the wait's bound expired, seconds later (profiles/r2c_concurrency_note.txt).  So the
// library's HIP calls on one device are serialised across threads by a recursive mutex, and a pipelined round holds it from the
// launch of its wait kernel until the challenge has been handed over (~ one round, tens of microseconds); everything else holds
// it only for the duration of its own calls.  (HIP calls made by OTHER code of the process on the same device during such a window
// can still delay a proof; the waits are bounded and the library then reports a void proof instead of a wrong one.)
// (A host transport's collective blocks until every rank has called it, and ranks may be threads that share the device: the gate is
// let go around those calls -- GateYield -- which is safe because such a communicator never has a pipelined round in flight.)
static std::recursive_mutex g_gate_mutex[64];
thread_local uint16_t g_gate_depth[64];
void gate_lock(int device) {
    g_gate_mutex[(unsigned)device & 63u].lock();
    ++g_gate_depth[(unsigned)device & 63u];
}
void gate_unlock(int device) {
    --g_gate_depth[(unsigned)device & 63u];
    g_gate_mutex[(unsigned)device & 63u].unlock();
}
void sc_internal_gate_lock(int device) { gate_lock(device); } // gkr.hip
void sc_internal_gate_unlock(int device) { gate_unlock(device); }

uint64_t sc_internal_cache_limit(); // sc_set_cache_limit: what each process-wide cache may keep (defined with the handle pool)
// shared with gkr.hip
int sc_internal_device() { return g_device; }
int &sc_internal_device_ref() { return g_device; } // the calling thread's device (sc_set_device), for gkr.hip

int sc_internal_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }
extern "C" const char *sc_last_error(void) { return g_last_error.c_str(); }
extern "C" int sc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
extern "C" int sc_set_device(int ordinal) {
    HIP_TRY(hipSetDevice(ordinal));
    g_device = ordinal;
    return SC_OK;
}

// ---------------------------------------------------------------------------------------------------
// transcript ABI (host)
// ---------------------------------------------------------------------------------------------------
extern "C" sc_rng *sc_rng_setup(void) { return new (std::nothrow) sc_rng(); }
extern "C" void sc_rng_free(sc_rng *rng) { delete rng; }
extern "C" void sc_rng_feed_bytes(sc_rng *rng, const uint8_t *buf, size_t len) { rng->rng.feed_bytes(buf, len); }
extern "C" void sc_rng_fill_bytes(sc_rng *rng, uint8_t *dest, size_t len) { rng->rng.fill_bytes(dest, len); }
extern "C" void sc_rng_feed_poly_info(sc_rng *rng, uint64_t max_multiplicands, uint64_t num_variables) {
    rng->rng.feed_poly_info(max_multiplicands, num_variables);
}
extern "C" void sc_rng_feed_prover_msg(sc_rng *rng, const uint64_t *evals, uint32_t n) {
    rng->rng.feed_prover_msg(reinterpret_cast<const sch::Fr *>(evals), n);
}
extern "C" void sc_rng_sample_fr(sc_rng *rng, uint64_t *out) {
    const sch::Fr a = rng->rng.sample_fr();
    std::memcpy(out, a.l, 32);
}

// ---------------------------------------------------------------------------------------------------
// ProverState in HBM

int resident_quiesce(sc_prover *p); // the interactive protocol's resident kernel leaves before anything else touches the handle
void prover_destroy(sc_prover *p) {
    if (!p) return;
    (void)resident_quiesce(p);
    DeviceGate gate_(p->device);
    (void)hipSetDevice(p->device);
    if (p->deferred_pending && p->sig) { // release a stream that still waits for a challenge before synchronising it
        __atomic_store_n(p->sig, p->sig_seq, __ATOMIC_RELEASE);
        p->deferred_pending = false;
    }
    if (p->own_stream) (void)hipStreamSynchronize(p->own_stream);
    if (p->arena) (void)hipFree(p->arena);
    if (p->d_partials) (void)hipFree(p->d_partials);
    if (p->d_partials2) (void)hipFree(p->d_partials2);
    if (p->d_fin_counters) (void)hipFree(p->d_fin_counters);
    if (p->d_finprods) (void)hipFree(p->d_finprods);
    if (p->d_W) (void)hipFree(p->d_W);
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    if (p->d_sums[0]) (void)hipFree(p->d_sums[0]);
    if (p->d_out) (void)hipFree(p->d_out);
    if (p->h_out) (void)hipHostFree(p->h_out);
    if (p->h_flag) (void)hipHostFree(p->h_flag);
    for (int q = 0; q < 2; ++q) {
        if (p->ring[q]) (void)hipFree(p->ring[q]);
        if (p->ev_copied[q]) (void)hipEventDestroy(p->ev_copied[q]);
        if (p->ev_consumed[q]) (void)hipEventDestroy(p->ev_consumed[q]);
    }
    if (p->d_chunk_msg) (void)hipFree(p->d_chunk_msg);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    if (p->copy_stream2) (void)hipStreamDestroy(p->copy_stream2);
    for (int q = 0; q < 2; ++q)
        if (p->ev_copied2[q]) (void)hipEventDestroy(p->ev_copied2[q]);
    if (p->tail) prover_destroy(p->tail);
    if (p->d_tail_send) (void)hipFree(p->d_tail_send);
    if (p->d_tail_recv) (void)hipFree(p->d_tail_recv);
    if (p->d_tail_tabs) (void)hipFree(p->d_tail_tabs);
    if (p->d_wide) (void)hipFree(p->d_wide);
    if (p->h_wide) (void)hipHostFree(p->h_wide);
    if (p->d_combos) (void)hipFree(p->d_combos);
    if (p->h_mail) (void)hipHostFree(p->h_mail);
    if (p->d_mail) (void)hipFree(p->d_mail);
    if (p->d_tail_sync) (void)hipFree(p->d_tail_sync);
    if (p->d_tail_xw) (void)hipFree(p->d_tail_xw);
    if (p->d_vram_mail) (void)hipFree(p->d_vram_mail);
    if (p->d_cur_tables) (void)hipFree(p->d_cur_tables);
    if (p->h_cur_tables) (void)hipHostFree(p->h_cur_tables);
    if (p->d_slot_table) (void)hipFree(p->d_slot_table);
    if (p->d_slot_exp) (void)hipFree(p->d_slot_exp);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    for (hipEvent_t e : p->prod_ev) (void)hipEventDestroy(e);
    if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    delete p;
}

bool handle_pool_offer(sc_prover *p);
extern "C" void sc_prover_free(sc_prover *p) {
    if (p) (void)resident_quiesce(p);
    if (p && handle_pool_offer(p)) return; // (a handle sc_ml_prove built: kept for the next proof of the same shape)
    prover_destroy(p);
}

int validate_desc(const sc_poly_desc *d) {
    if (!d) return sc_internal_fail(SC_ERR_BAD_ARG, "null descriptor");
    if (d->num_vars == 0) return sc_internal_fail(SC_ERR_CONSTANT_POLY, "Attempt to prove a constant.");
    if (d->num_vars > 40) return sc_internal_fail(SC_ERR_BAD_ARG, "num_vars %u too large", d->num_vars);
    if (d->n_tables == 0 || !d->tables) return sc_internal_fail(SC_ERR_BAD_ARG, "no tables");
    if (d->n_products && (!d->coeffs || !d->prod_offsets || !d->prod_indices)) return sc_internal_fail(SC_ERR_BAD_ARG, "null product arrays");
    uint32_t mx = 0;
    for (uint32_t k = 0; k < d->n_products; ++k) {
        if (d->prod_offsets[k + 1] <= d->prod_offsets[k]) return sc_internal_fail(SC_ERR_BAD_ARG, "product %u is empty", k); // data_structures.rs:78
        mx = std::max(mx, d->prod_offsets[k + 1] - d->prod_offsets[k]);
        for (uint32_t q = d->prod_offsets[k]; q < d->prod_offsets[k + 1]; ++q)
            if (d->prod_indices[q] >= d->n_tables) return sc_internal_fail(SC_ERR_BAD_ARG, "product %u refers to table %u >= %u", k, d->prod_indices[q], d->n_tables);
    }
    if (mx != d->max_multiplicands) return sc_internal_fail(SC_ERR_BAD_ARG, "max_multiplicands %u != max product length %u", d->max_multiplicands, mx);
    for (uint32_t u = 0; u < d->n_tables; ++u)
        if (!d->tables[u]) return sc_internal_fail(SC_ERR_BAD_ARG, "table %u is null", u);
    return SC_OK;
}

sch::Fr fr_small(int64_t v) { return v >= 0 ? sch::from_u64((uint64_t)v) : sch::neg(sch::from_u64((uint64_t)(-v))); }

// (deg+1) x (M+1) matrix taking a degree-M polynomial's values at the kernel nodes (scd::node_value: 0, 1, inf, -1, 2, ...)
// to its values at 0..deg, times `scale`.  Exact Lagrange weights in the field; "inf" is the leading coefficient L:
// P(t) = L t^M + sum_i (P(x_i) - L x_i^M) l_i(t) over the M finite nodes.
void build_node_matrix(uint32_t M, uint32_t D, const sch::Fr &scale, std::vector<sch::Fr> &out) {
    std::vector<int64_t> xs;
    std::vector<uint32_t> fin;
    int inf_col = -1;
    for (uint32_t s = 0; s <= M; ++s) {
        const int32_t nv = scd::node_value((int)s);
        if (nv == scd::kNodeInf) inf_col = (int)s;
        else fin.push_back(s);
        xs.push_back(nv);
    }
    out.assign((size_t)D * (M + 1), sch::zero());
    for (uint32_t t = 0; t < D; ++t) {
        sch::Fr inf_w = sch::kOne; // t^M
        for (uint32_t e = 0; e < M; ++e) inf_w = sch::mul(inf_w, fr_small(t));
        for (uint32_t s : fin) {
            sch::Fr num = sch::kOne, den = sch::kOne;
            for (uint32_t j : fin) {
                if (j == s) continue;
                num = sch::mul(num, fr_small((int64_t)t - xs[j]));
                den = sch::mul(den, fr_small(xs[s] - xs[j]));
            }
            const sch::Fr l = sch::mul(num, sch::inverse(den));
            out[(size_t)t * (M + 1) + s] = sch::mul(scale, l);
            if (inf_col >= 0) {
                sch::Fr xm = sch::kOne;
                for (uint32_t e = 0; e < M; ++e) xm = sch::mul(xm, fr_small(xs[s]));
                inf_w = sch::sub(inf_w, sch::mul(xm, l));
            }
        }
        if (inf_col >= 0) out[(size_t)t * (M + 1) + inf_col] = sch::mul(scale, inf_w);
    }
}

// The weights of a degree-M polynomial's values at the kernel nodes (0, 1, inf, -1, 2) in its value at an arbitrary point r:
// lam[s] for s = 0..M, M <= 4.  The denominators' inverses are computed once; a call is ~20 field products (it runs on the host while
// the round kernel runs on the device).
void claim_weights(uint32_t M, const sch::Fr &r, sch::Fr *lam) {
    struct Den {
        sch::Fr inv[5][5]; // inv[M][s]: 1 / prod_{j != s, finite}(x_s - x_j)
        Den() {
            for (uint32_t m = 1; m <= 4; ++m)
                for (uint32_t s = 0; s <= m; ++s) {
                    inv[m][s] = sch::zero();
                    if (scd::node_value((int)s) == scd::kNodeInf) continue;
                    sch::Fr den = sch::kOne;
                    for (uint32_t j = 0; j <= m; ++j)
                        if (j != s && scd::node_value((int)j) != scd::kNodeInf) den = sch::mul(den, fr_small((int64_t)scd::node_value((int)s) - scd::node_value((int)j)));
                    inv[m][s] = sch::inverse(den);
                }
        }
    };
    static const Den den;
    sch::Fr diff[5]; // r - x_j
    for (uint32_t j = 0; j <= M; ++j)
        if (scd::node_value((int)j) != scd::kNodeInf) diff[j] = sch::sub(r, fr_small(scd::node_value((int)j)));
    sch::Fr inf_w = sch::kOne; // r^M - sum_s x_s^M l_s(r)
    for (uint32_t e = 0; e < M; ++e) inf_w = sch::mul(inf_w, r);
    int inf_col = -1;
    for (uint32_t s = 0; s <= M; ++s) {
        const int32_t xs = scd::node_value((int)s);
        if (xs == scd::kNodeInf) {
            inf_col = (int)s;
            continue;
        }
        sch::Fr l = den.inv[M][s];
        for (uint32_t j = 0; j <= M; ++j)
            if (j != s && scd::node_value((int)j) != scd::kNodeInf) l = sch::mul(l, diff[j]);
        lam[s] = l;
        sch::Fr xm = sch::kOne;
        for (uint32_t e = 0; e < M; ++e) xm = sch::mul(xm, fr_small(xs));
        inf_w = sch::sub(inf_w, sch::mul(xm, l));
    }
    if (inf_col >= 0) lam[inf_col] = inf_w;
}

int prover_build(const sc_poly_desc *d, sc_prover *p) {
    DeviceGate gate_(g_device);
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    p->device = g_device;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking));
    p->stream = p->own_stream;
    HIP_TRY(hipEventCreate(&p->ev0));
    HIP_TRY(hipEventCreate(&p->ev1));
    p->nv = d->num_vars;
    p->max_mult = d->max_multiplicands;
    p->D = d->max_multiplicands + 1;
    p->K = d->n_products;
    p->U = d->n_tables;
    p->randomness.reserve(p->nv);
    if (d->flags & SC_NO_DEVICE_POLLING) {
        p->pipeline_ok = false;
        p->polling_off_by_caller = true;
    }
#ifdef SC_EXPERIMENTS
    if (const char *e = std::getenv("SC_FE")) p->use_fe = std::atoi(e) != 0;
    if (const char *e = std::getenv("SC_KERNEL")) p->kernel_variant = std::atoi(e);
    // one arithmetic per round: k_finalize's 2^(5(M-1)) compensation is chosen per round, so the saturated kernels never share a
    // round with the (carry-free) tree kernel
    if (!p->use_fe && p->kernel_variant == 3) p->kernel_variant = 0;
#endif
    p->use_f29 = p->kernel_variant == 3;
    p->wide_tree = wide_tree_enabled(); // (the policy as it is NOW: the handle's table format depends on it)
#ifdef SC_EXPERIMENTS
    if (const char *e = std::getenv("SC_F29")) p->use_f29 = p->use_f29 && std::atoi(e) != 0;
#endif
    // (every big-round kernel of the handle must read the format: the tree kernels do, for up to eight factors with kernels_wide.hip)
    for (uint32_t k = 0; k < d->n_products; ++k)
        if (d->prod_offsets[k + 1] - d->prod_offsets[k] > (p->wide_tree ? 8u : 4u)) p->use_f29 = false;
    // with more tables than the small-round kernels take, the big-round kernels also run the short rounds, whose tables are
    // smaller than one 128-entry block of the chunk-planar layout
    if (d->n_tables > (uint32_t)scd::kMaxSmallTables) p->use_f29 = false;
    p->merge_rounds = p->kernel_variant == 3 && d->n_products > 0 && d->n_products <= (uint32_t)scd::kMaxRoundProds;
    for (uint32_t k = 0; k < d->n_products; ++k)
        if (d->prod_offsets[k + 1] - d->prod_offsets[k] > 4) p->merge_rounds = false;
#ifdef SC_EXPERIMENTS
    if (const char *e = std::getenv("SC_MERGE")) p->merge_rounds = p->merge_rounds && std::atoi(e) != 0;
    if (const char *e = std::getenv("SC_FUSED_FIN")) p->fused_finalize = std::atoi(e) != 0;
#endif
    p->use_tail = scd::policy(scd::kPolTail) != 0; // 0: late rounds as pipelined launches (the path sharded RCCL proofs take)

    // products: distinct tables + multiplicities
    uint64_t partial_elems = 0;
    std::vector<uint32_t> slot_table, slot_exp;
    std::vector<FinProd> fin(p->K);
    std::vector<Combo> combos;
    std::vector<sch::Fr> Wall;
    for (uint32_t k = 0; k < p->K; ++k) {
        Product pr;
        std::memcpy(&pr.coeff, d->coeffs + 4 * k, 32);
        if (sch::geq_p(pr.coeff)) return sc_internal_fail(SC_ERR_BAD_ARG, "coefficient %u is not a canonical field element", k);
        for (uint32_t q = d->prod_offsets[k]; q < d->prod_offsets[k + 1]; ++q) {
            const uint32_t t = d->prod_indices[q];
            auto it = std::find(pr.tables.begin(), pr.tables.end(), t);
            if (it == pr.tables.end()) {
                pr.tables.push_back(t);
                pr.exps.push_back(1);
            } else {
                pr.exps[it - pr.tables.begin()]++;
            }
        }
        p->prod_indices.emplace_back(d->prod_indices + d->prod_offsets[k], d->prod_indices + d->prod_offsets[k + 1]);
        pr.M = d->prod_offsets[k + 1] - d->prod_offsets[k];
        pr.fused = pr.M <= (uint32_t)scd::kMaxFusedM;
        if (!pr.fused) p->any_generic = true;
        pr.partial_off = partial_elems;
        partial_elems += (uint64_t)scd::kMaxGrid * (pr.M + 1);
        pr.slot_off = (uint32_t)slot_table.size();
        slot_table.insert(slot_table.end(), pr.tables.begin(), pr.tables.end());
        slot_exp.insert(slot_exp.end(), pr.exps.begin(), pr.exps.end());
        fin[k].M = pr.M;
        fin[k].pad = 0;
        fin[k].partial_off = pr.partial_off;
        {
            sch::Fr sc = pr.coeff; // coeff * 2^(5(M-1)) in Montgomery form = Montgomery form doubled 5(M-1) times
            for (uint32_t dbl = 0; dbl < 5 * (pr.M - 1); ++dbl) sc = sch::add(sc, sc);
            std::vector<sch::Fr> w;
            fin[k].w_off = Wall.size();
            build_node_matrix(pr.M, p->D, pr.coeff, w);
            Wall.insert(Wall.end(), w.begin(), w.end());
            build_node_matrix(pr.M, p->D, sc, w);
            Wall.insert(Wall.end(), w.begin(), w.end());
        }
        for (uint32_t t = 0; t <= pr.M; ++t) {
            Combo c;
            c.t = t;
            c.M = pr.M;
            c.slot_off = pr.slot_off;
            c.n_slots = (uint32_t)pr.tables.size();
            c.partial_off = pr.partial_off;
            combos.push_back(c);
        }
        p->prods.push_back(std::move(pr));
    }

    // table memory: copy mode = A (2^nv) + B (2^(nv-1)); borrow mode = B (2^(nv-1)) + C (2^(nv-2))
    const bool on_device = d->flags & SC_TABLES_ON_DEVICE;
    const bool borrow = on_device && (d->flags & SC_TABLES_BORROW);
    const uint64_t n = 1ULL << p->nv;
    // streamed: only where it can matter (>= 2^11 entries) and where the merged big-round kernel applies (it is what walks the chunks)
    const bool streamed = !on_device && (d->flags & SC_TABLES_STREAM) && p->nv >= 11;
    const bool small_foot = borrow || streamed; // the caller's tables are only read: the handle holds the bound tables alone
    bool staged = false;
    const uint64_t s0 = small_foot ? std::max<uint64_t>(n >> 1, 1) : n;
    const uint64_t s1 = small_foot ? std::max<uint64_t>(n >> 2, 1) : std::max<uint64_t>(n >> 1, 1);
    const uint64_t per_table = (s0 + s1) * 36; // 32 B main + 4 B limb-8 array per element (internal F29 format)
    HIP_TRY(hipMalloc(&p->arena, per_table * p->U));
    p->arena_bytes = per_table * p->U;
    p->tabs.resize(p->U);
    p->borrow = borrow;
    // Device tables are copied on the handle's own non-blocking stream, which is ordered after nothing the caller enqueued:
    // wait for whatever produced them (any stream of this device) before reading.
    if (on_device && !borrow) HIP_TRY(hipDeviceSynchronize());
    for (uint32_t u = 0; u < p->U; ++u) {
        Table &t = p->tabs[u];
        char *base = static_cast<char *>(p->arena) + per_table * u;
        t.buf[0] = reinterpret_cast<uint4 *>(base);
        t.buf[1] = reinterpret_cast<uint4 *>(base + s0 * 32);
        t.buf_top[0] = reinterpret_cast<int32_t *>(base + (s0 + s1) * 32);
        t.buf_top[1] = t.buf_top[0] + s0;
        if (streamed) {
            t.cur = nullptr; // nothing resident before round 2
            t.next = 0;
            p->host_tabs.push_back(d->tables[u]);
        } else if (borrow) {
            t.cur = reinterpret_cast<const uint4 *>(d->tables[u]);
            t.next = 0;
            p->origin.push_back(t.cur);
        } else {
            if (!on_device && (d->flags & SC_TABLES_STREAM)) p->host_tabs.push_back(d->tables[u]); // too small to stream: copied, but rewound like a streamed handle
            // (host tables of a shape the merged big-round kernel takes: copied at the END of the build, in chunks, with round 1 computed under the copy)
            if (!(staged = !on_device && staged_init_applies(p)))
                HIP_TRY(hipMemcpyAsync(t.buf[0], d->tables[u], n * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, p->stream));
            t.cur = t.buf[0];
            t.next = 1;
        }
    }

    if (streamed) {
        p->streamed = true;
        uint32_t cl = p->stream_chunk_request ? p->stream_chunk_request : 22u; // 2^22 entries = 128 MiB per table and chunk
        cl = std::max(10u, std::min(cl, p->nv));                               // >= 2^10: the chunk-planar F29 blocks of the bound half stay aligned
        p->chunk_log2 = cl;
        for (int q = 0; q < 2; ++q) {
            HIP_TRY(hipMalloc(&p->ring[q], ((size_t)p->U << cl) * 32));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_copied[q], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_consumed[q], hipEventDisableTiming));
        }
        HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipMalloc(&p->d_chunk_msg, (size_t)2 * p->D * 32));
    }
    HIP_TRY(hipMalloc(&p->d_partials, std::max<uint64_t>(partial_elems, 1) * 32));
    HIP_TRY(hipMalloc(&p->d_partials2, std::max<uint64_t>(partial_elems, 1) * 32)); // second level of the in-kernel finalize (k_round_tree)
    HIP_TRY(hipMalloc(&p->d_fin_counters, 4 * (2 + scd::kMaxGrid / 32 + 16)));
    HIP_TRY(hipMemsetAsync(p->d_fin_counters, 0, 4 * (2 + scd::kMaxGrid / 32 + 16), p->stream));
    p->d_fin_mb_counter = p->d_fin_counters + (2 + scd::kMaxGrid / 32 + 8); // k_finalize_mb's arrival counter
    HIP_TRY(hipMalloc(&p->d_finprods, std::max<size_t>(p->K, 1) * sizeof(FinProd)));
    if (p->K) HIP_TRY(hipMemcpyAsync(p->d_finprods, fin.data(), p->K * sizeof(FinProd), hipMemcpyHostToDevice, p->stream));
    p->h_finprods = fin;
    // twice: the matrices the kernels read, and behind them the copy sc_internal_scale_by_bound_table multiplies from (and a reset restores)
    HIP_TRY(hipMalloc(&p->d_W, std::max<size_t>(2 * Wall.size(), 1) * 32));
    p->w_elems = (uint32_t)Wall.size();
    if (!Wall.empty()) {
        HIP_TRY(hipMemcpyAsync(p->d_W, Wall.data(), Wall.size() * 32, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(p->d_W + Wall.size(), Wall.data(), Wall.size() * 32, hipMemcpyHostToDevice, p->stream));
    }
    HIP_TRY(hipMalloc(&p->d_scratch, (size_t)(2 + p->D) * std::max<uint32_t>(p->K, 1) * p->D * 32));
    {
        const size_t one = (size_t)std::max<uint32_t>(p->K, 1) * p->D * 32;
        HIP_TRY(hipMalloc(&p->d_sums[0], 2 * one));
        p->d_sums[1] = reinterpret_cast<FrHost *>(reinterpret_cast<char *>(p->d_sums[0]) + one);
    }
    HIP_TRY(hipMalloc(&p->d_out, (size_t)p->D * 32));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_out), (size_t)p->D * 32, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_flag), 64, hipHostMallocMapped | hipHostMallocCoherent));
    *p->h_flag = 0;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_out_dev), p->h_out, 0));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_flag_dev), p->h_flag, 0));
    p->has_meta = combos.size() <= (size_t)scd::kMetaCombos && slot_table.size() <= (size_t)scd::kMetaSlots;
    if (p->has_meta) {
        std::memset(&p->meta, 0, sizeof(p->meta));
        std::copy(combos.begin(), combos.end(), p->meta.combo);
        std::copy(slot_table.begin(), slot_table.end(), p->meta.slot_table);
        std::copy(slot_exp.begin(), slot_exp.end(), p->meta.slot_exp);
    }
    p->n_combos = (int)combos.size();
    HIP_TRY(hipMalloc(&p->d_combos, std::max<size_t>(combos.size(), 1) * sizeof(Combo)));
    if (!combos.empty()) HIP_TRY(hipMemcpyAsync(p->d_combos, combos.data(), combos.size() * sizeof(Combo), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMalloc(&p->d_slot_table, std::max<size_t>(slot_table.size(), 1) * 4));
    HIP_TRY(hipMalloc(&p->d_slot_exp, std::max<size_t>(slot_exp.size(), 1) * 4));
    if (!slot_table.empty()) {
        HIP_TRY(hipMemcpyAsync(p->d_slot_table, slot_table.data(), slot_table.size() * 4, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(p->d_slot_exp, slot_exp.data(), slot_exp.size() * 4, hipMemcpyHostToDevice, p->stream));
    }
    if (p->any_generic || p->U > (uint32_t)scd::kMaxSmallTables) { // (two sets: a streamed handle's chunks alternate between them, as between the staging slots)
        HIP_TRY(hipMalloc(&p->d_cur_tables, 2 * p->U * sizeof(void *)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_cur_tables), 2 * p->U * sizeof(void *), hipHostMallocDefault));
    }
    if (staged) {
        int rc_s = staged_copy_and_round1(p, d->tables);
        if (rc_s) return rc_s;
    }
    HIP_TRY(hipStreamSynchronize(p->stream)); // inputs are copied: the caller may drop them now (prover.rs:55-59)
    return SC_OK;
}

std::vector<uint8_t> pool_key_of(const sc_poly_desc *d, int device);
sc_prover *handle_pool_take(const std::vector<uint8_t> &key);
// Per-owner policy of a handle, as prover_build leaves it: whether kernels may wait for the host (SC_NO_DEVICE_POLLING /
// sc_prover_set_polling), the resident kernel's patience (sc_prover_set_resident), per-launch timing (sc_prover_set_timing).  A handle
// that comes back from the pool starts from these, whatever its previous owner had set.
void reset_owner_policy(sc_prover *p, uint32_t desc_flags) {
    p->polling_off_by_caller = (desc_flags & SC_NO_DEVICE_POLLING) != 0;
    p->pipeline_ok = !p->polling_off_by_caller;
    p->resident_spins = kResidentSpinsDefault;
    if (p->timing) (void)sc_prover_set_timing(p, 0);
    p->n_retries = 0;
}
extern "C" int sc_prover_init(const sc_poly_desc *desc, sc_prover **out) {
    if (!out) return sc_internal_fail(SC_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    // a freed prover of the same structure on this device, if the pool has one (see handle_pool_*: built once, rewound afterwards)
    std::vector<uint8_t> key = pool_key_of(desc, g_device);
    if (sc_prover *kept = handle_pool_take(key)) {
        if (sc_prover_reset(kept, desc->tables, desc->flags & SC_TABLES_ON_DEVICE) == SC_OK) {
            reset_owner_policy(kept, desc->flags); // (nothing a previous owner set -- polling, the resident kernel's patience -- carries over)
            *out = kept;
            return SC_OK;
        }
        kept->pool_key.clear();
        prover_destroy(kept);
    }
    sc_prover *p = new (std::nothrow) sc_prover();
    if (!p) return sc_internal_fail(SC_ERR_OOM, "host allocation failed");
    rc = prover_build(desc, p);
    if (rc) {
        prover_destroy(p);
        return rc;
    }
    p->pool_key = std::move(key);
    *out = p;
    return SC_OK;
}

extern "C" int sc_prover_init_streamed(const sc_poly_desc *desc, uint32_t chunk_log2, sc_prover **out) {
    if (!out) return sc_internal_fail(SC_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (desc->flags & SC_TABLES_ON_DEVICE) return sc_internal_fail(SC_ERR_BAD_ARG, "streamed tables are host tables");
    sc_prover *p = new (std::nothrow) sc_prover();
    if (!p) return sc_internal_fail(SC_ERR_OOM, "host allocation failed");
    p->stream_chunk_request = chunk_log2;
    sc_poly_desc d2 = *desc;
    d2.flags |= SC_TABLES_STREAM;
    rc = prover_build(&d2, p);
    if (rc) {
        prover_destroy(p);
        return rc;
    }
    *out = p;
    return SC_OK;
}

extern "C" int sc_prover_set_stream(sc_prover *p, void *hip_stream, int use_own) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    DeviceGate gate_(p->device);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->stream = use_own ? p->own_stream : static_cast<hipStream_t>(hip_stream); // NULL = the legacy default stream
    return SC_OK;
}

// fold the previous round's event pairs into the accumulators (blocks until that round has finished)
int collect_timing(sc_prover *p) {
    if (!p->timing || !p->timing_pending) return SC_OK;
    HIP_TRY(hipEventSynchronize(p->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->rounds_ms += ms;
    for (uint32_t k = 0; k < (p->prod_merged ? 1u : p->K) && p->prod_timed; ++k) { // big rounds only: one fused kernel launch per product
        HIP_TRY(hipEventElapsedTime(&ms, p->prod_ev[2 * k], p->prod_ev[2 * k + 1]));
        p->prod_ms[k] += ms;
        p->prod_launches[k] += 1;
        if (p->timed_round >= 1 && p->timed_round <= p->round_kernel_ms.size()) { // (per-product launches of one round add up)
            p->round_kernel_ms[p->timed_round - 1] += ms;
            if (k == 0) p->round_kernel_launches[p->timed_round - 1] += 1;
        }
    }
    p->timing_pending = false;
    return SC_OK;
}

// Bind the challenge `r` into every table once more and write the results back to back (table u at d_out + u * n * 4 limbs, n =
// 2^(num_vars - round) entries each, canonical form whatever the tables' internal format).  After this the handle is exhausted.
int prover_bind_out(sc_prover *p, const uint64_t *r, uint64_t *d_out) {
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    DeviceGate gate_(p->device);
    if (p->exhausted || p->round == 0 || p->round > p->nv) return sc_internal_fail(SC_ERR_NOT_ACTIVE, "bind needs a prover that has run at least one round");
    if (p->streamed && p->round < 2) return sc_internal_fail(SC_ERR_NOT_ACTIVE, "a streamed handle's tables are resident from round 2 on: nothing to bind yet");
    sch::Fr rr;
    std::memcpy(&rr, r, 32);
    if (sch::geq_p(rr)) return sc_internal_fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
    HIP_TRY(hipSetDevice(p->device));
    p->randomness.push_back(rr);
    const uint64_t n_out = 1ULL << (p->nv - p->round);
    for (uint32_t u0 = 0; u0 < p->U; u0 += (uint32_t)scd::kMaxSmallTables) { // one launch per 32 tables
        const uint32_t cnt = std::min<uint32_t>(p->U - u0, (uint32_t)scd::kMaxSmallTables);
        TablePtrs tp;
        std::memset(&tp, 0, sizeof(tp));
        for (uint32_t j = 0; j < cnt; ++j) {
            tp.src[j] = p->tabs[u0 + j].cur;
            tp.src_top[j] = p->tabs[u0 + j].cur_top;
            tp.dst[j] = reinterpret_cast<uint4 *>(d_out + 4 * n_out * (size_t)(u0 + j));
        }
        HIP_TRY(scd::launch_fix_multi(tp, (int)cnt, to_dev(rr), nullptr, n_out, p->stream));
    }
    p->exhausted = true;
    return SC_OK;
}

extern "C" int sc_prover_bind_final(sc_prover *p, const uint64_t *r, uint64_t *d_out) {
    if (!p || !r || !d_out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (p->exhausted || p->round != p->nv) return sc_internal_fail(SC_ERR_NOT_ACTIVE, "bind_final needs a prover that has finished its last local round");
    return prover_bind_out(p, r, d_out);
}

extern "C" int sc_prover_push_randomness(sc_prover *p, const uint64_t *r) {
    if (!p || !r) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    sch::Fr rr;
    std::memcpy(&rr, r, 32);
    p->randomness.push_back(rr);
    return SC_OK;
}

extern "C" int sc_prover_state(sc_prover *p, uint64_t *randomness, uint32_t *n_randomness, uint64_t *tables_out, uint32_t *round) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    if (tables_out) {
        int rc_q = resident_quiesce(p);
        if (rc_q) return rc_q;
    }
    DeviceGate gate_(p->device);
    if (randomness && !p->randomness.empty()) std::memcpy(randomness, p->randomness.data(), p->randomness.size() * 32);
    if (n_randomness) *n_randomness = (uint32_t)p->randomness.size();
    if (round) *round = p->round;
    if (tables_out) {
        if (p->exhausted) return sc_internal_fail(SC_ERR_NOT_ACTIVE, "tables were consumed by sc_prover_bind_final");
        HIP_TRY(hipSetDevice(p->device));
        const uint32_t bound = p->round > 0 ? p->round - 1 : 0;
        const uint64_t n = 1ULL << (p->nv - bound);
        if (p->streamed && p->round < 2) { // nothing bound yet: the tables are the caller's host arrays
            for (uint32_t u = 0; u < p->U; ++u) std::memcpy(tables_out + 4 * n * u, p->host_tabs[u], n * 32);
            return SC_OK;
        }
        void *tmp = nullptr; // tables in the internal F29 format are converted to the canonical reference layout first
        for (uint32_t u = 0; u < p->U; ++u) {
            const void *src = p->tabs[u].cur;
            if (p->tabs[u].cur_top) {
                if (!tmp) HIP_TRY(hipMalloc(&tmp, n * 32));
                HIP_TRY(scd::launch_f29_to_sat(p->tabs[u].cur, p->tabs[u].cur_top, static_cast<uint4 *>(tmp), n, p->stream));
                src = tmp;
            }
            HIP_TRY(hipMemcpyAsync(tables_out + 4 * n * u, src, n * 32, hipMemcpyDeviceToHost, p->stream));
            if (tmp) HIP_TRY(hipStreamSynchronize(p->stream));
        }
        if (tmp) (void)hipFree(tmp);
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    return SC_OK;
}

extern "C" int sc_prover_last_round_ms(sc_prover *p, float *ms) {
    if (!p || !ms) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (!p->timed) return sc_internal_fail(SC_ERR_BAD_ARG, "no timed round: enable sc_prover_set_timing before the round");
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(ms, p->ev0, p->ev1));
    return SC_OK;
}

extern "C" int sc_prover_set_timing(sc_prover *p, int on) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    DeviceGate gate_(p->device);
    HIP_TRY(hipSetDevice(p->device));
    if (on && p->prod_ev.empty()) {
        p->prod_ev.resize(2 * (size_t)p->K);
        for (auto &e : p->prod_ev) HIP_TRY(hipEventCreate(&e));
    }
    p->timing = on != 0;
    p->timing_pending = false;
    p->prod_ms.assign(p->K, 0.0);
    p->prod_launches.assign(p->K, 0);
    p->rounds_ms = 0.0;
    p->round_kernel_ms.assign(p->nv, 0.0);
    p->round_kernel_launches.assign(p->nv, 0);
    return SC_OK;
}

// per round (index = round - 1, p->nv entries): accumulated device time of the big-round kernel launch(es) of that round since
// sc_prover_set_timing(p, 1), and the number of timed proofs that contributed (0 for the latency-bound rounds, which record no events)
extern "C" int sc_prover_get_round_timing(sc_prover *p, double *ms_per_round, uint64_t *launches_per_round) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    DeviceGate gate_(p->device);
    if (!p->timing) return sc_internal_fail(SC_ERR_BAD_ARG, "timing is not enabled on this handle");
    HIP_TRY(hipSetDevice(p->device));
    int rc = collect_timing(p);
    if (rc) return rc;
    for (uint32_t i = 0; i < p->nv; ++i) {
        if (ms_per_round) ms_per_round[i] = i < p->round_kernel_ms.size() ? p->round_kernel_ms[i] : 0.0;
        if (launches_per_round) launches_per_round[i] = i < p->round_kernel_launches.size() ? p->round_kernel_launches[i] : 0;
    }
    return SC_OK;
}

extern "C" int sc_prover_get_timing(sc_prover *p, double *ms_per_product, uint64_t *launches_per_product, double *rounds_ms) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    DeviceGate gate_(p->device);
    if (!p->timing) return sc_internal_fail(SC_ERR_BAD_ARG, "timing is not enabled on this handle");
    HIP_TRY(hipSetDevice(p->device));
    int rc = collect_timing(p);
    if (rc) return rc;
    for (uint32_t k = 0; k < p->K; ++k) {
        if (ms_per_product) ms_per_product[k] = p->prod_ms[k];
        if (launches_per_product) launches_per_product[k] = p->prod_launches[k];
    }
    if (rounds_ms) *rounds_ms = p->rounds_ms;
    return SC_OK;
}

// Rewind a handle to round 0 without reallocating.  Borrow mode: tables_or_null = new borrowed device
// pointers (NULL = the same tables again).  Copy mode: tables must be given and are copied in again
// (host pointers, or device pointers when flags has SC_TABLES_ON_DEVICE).
extern "C" int sc_prover_reset(sc_prover *p, const uint64_t *const *tables_or_null, uint32_t flags) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    (void)resident_quiesce(p); // (a failed resident kernel leaves the handle exhausted: exactly what a reset repairs)
    DeviceGate gate_(p->device);
    HIP_TRY(hipSetDevice(p->device));
    abandon_deferred(p);
    if (wait_gave_up(p)) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        __atomic_store_n(p->sig + 1, 0u, __ATOMIC_RELEASE);
    }
    {
        int rc_t = collect_timing(p);
        if (rc_t) return rc_t;
    }
    p->r1_cached = false;
    if (p->w_scaled) { // (sc_internal_scale_by_bound_table: the polynomial is the descriptor's again)
        HIP_TRY(hipMemcpyAsync(p->d_W, p->d_W + p->w_elems, (size_t)p->w_elems * 32, hipMemcpyDeviceToDevice, p->stream));
        p->w_scaled = false;
    }
    const uint64_t n = 1ULL << p->nv;
    if (p->streamed) {
        if (flags & SC_TABLES_ON_DEVICE) return sc_internal_fail(SC_ERR_BAD_ARG, "streamed tables are host tables");
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (uint32_t u = 0; u < p->U; ++u) {
            if (tables_or_null) {
                if (!tables_or_null[u]) return sc_internal_fail(SC_ERR_BAD_ARG, "table %u is null", u);
                p->host_tabs[u] = tables_or_null[u];
            }
            p->tabs[u].cur = nullptr;
            p->tabs[u].cur_top = nullptr;
            p->tabs[u].next = 0;
        }
    } else if (p->borrow) {
        for (uint32_t u = 0; u < p->U; ++u) {
            if (tables_or_null) {
                if (!tables_or_null[u]) return sc_internal_fail(SC_ERR_BAD_ARG, "table %u is null", u);
                p->origin[u] = reinterpret_cast<const uint4 *>(tables_or_null[u]);
            }
            p->tabs[u].cur = p->origin[u];
            p->tabs[u].cur_top = nullptr;
            p->tabs[u].next = 0;
        }
    } else {
        if (!tables_or_null && p->host_tabs.size() == p->U) tables_or_null = p->host_tabs.data(); // sc_prover_init_streamed below its threshold
        if (!tables_or_null) return sc_internal_fail(SC_ERR_BAD_ARG, "a copying handle needs the tables again to reset");
        const bool on_device = flags & SC_TABLES_ON_DEVICE;
        if (on_device) HIP_TRY(hipDeviceSynchronize()); // the producer of the new tables may still be running on another stream
        const bool staged = !on_device && staged_init_applies(p);
        for (uint32_t u = 0; u < p->U; ++u) {
            if (!tables_or_null[u]) return sc_internal_fail(SC_ERR_BAD_ARG, "table %u is null", u);
            if (!staged)
                HIP_TRY(hipMemcpyAsync(p->tabs[u].buf[0], tables_or_null[u], n * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                       p->stream));
            p->tabs[u].cur = p->tabs[u].buf[0];
            p->tabs[u].cur_top = nullptr;
            p->tabs[u].next = 1;
        }
        if (staged) {
            p->round = 0; // (what staged_copy_and_round1 expects: a handle in front of its first round)
            p->exhausted = false;
            int rc_s = staged_copy_and_round1(p, tables_or_null);
            if (rc_s) return rc_s;
        }
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    p->round = 0;
    p->sums_round = -1;
    p->exhausted = false;
    p->randomness.clear();
    return SC_OK;
}

// ---- gkr.hip: phase two without the f3 pass ---------------------------------------------------------------------------------------
// GKRRoundSumcheck::prove multiplies f3 by the scalar f2(u) before phase two (gkr_round_sumcheck/mod.rs:71-75,122).  Phase one's prover
// has bound f2 at u_0..u_{dim-2} already (prover.rs:84-89): f2(u) = lo + u_last (hi - lo) of its final two-entry table, and a scalar
// factor of a one-product polynomial is that product's coefficient -- exact field arithmetic, the same canonical messages.  So:
//   sc_internal_bound_table(p, u)   the two-entry table the completed rounds left (device memory, canonical), to be taken BEFORE a reset
//   sc_internal_scale_by_bound_table(p, table, r_last)   after the reset, on the handle's stream: every node->message matrix becomes
//                                   (descriptor's matrix) x table(r_last); nothing visits the host.  The next reset restores the matrices.
hipStream_t sc_internal_prover_stream(sc_prover *p) { return p->stream; }
const void *sc_internal_bound_table(sc_prover *p, uint32_t u) {
    if (!p || u >= p->U || p->round != p->nv || p->exhausted || p->tabs[u].cur_top || p->res.active) return nullptr;
    return p->tabs[u].cur;
}
int sc_internal_scale_by_bound_table(sc_prover *p, const void *table, const sch::Fr &r_last) {
    if (!p || !table || p->round != 0) return sc_internal_fail(SC_ERR_BAD_ARG, "scale_by_bound_table: a freshly reset handle and a bound table");
    DeviceGate gate_(p->device);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(scd::launch_scale_w_by_table_eval(p->d_W, p->d_W + p->w_elems, p->w_elems, table, to_dev(r_last), p->stream));
    p->w_scaled = true;
    return SC_OK;
}

// ---------------------------------------------------------------------------------------------------
// DenseMultilinearExtension::fix_variables (below, next to evaluate: both are passes of k_fold_multi) and
// ListOfProductsOfPolynomials::evaluate (data_structures.rs:99-109): sum_k c_k prod_j T_j(point).
// The U table evaluations run on the device, three variables per pass (kernels.h: FoldArgs); the K + sum m_k
// scalar products that combine them are host work.
// ---------------------------------------------------------------------------------------------------
namespace {
struct DevMem { // frees on scope exit
    void *p = nullptr;
    ~DevMem() {
        if (p) (void)hipFree(p);
    }
};
struct StreamGuard {
    hipStream_t s = nullptr;
    ~StreamGuard() {
        if (s) (void)hipStreamDestroy(s);
    }
};
// sc_poly_evaluate's work areas and stream, kept between calls: three hipMalloc / hipFree pairs and a stream cost ~0.8 ms per call,
// half of an evaluation at 2^24 entries and most of one at 2^20.  One call at a time holds the lease; a concurrent call (another
// thread) allocates for itself.  sc_release_caches frees it.
struct EvalCache {
    std::mutex mu;
    void *buf = nullptr;
    size_t cap = 0;
    int device = -1;
    hipStream_t s = nullptr;
};
EvalCache g_eval_cache;
struct EvalLease { // RAII: the cache if it is free, nothing otherwise
    bool held = false;
    EvalLease() : held(g_eval_cache.mu.try_lock()) {}
    ~EvalLease() {
        if (held) g_eval_cache.mu.unlock();
    }
    // a buffer of at least `bytes` and a stream on `device`, or null (the caller then allocates)
    void *get(int device, size_t bytes, hipStream_t *s_out) {
        if (!held || bytes > sc_internal_cache_limit()) return nullptr; // (over the limit: the caller allocates and frees its own)
        EvalCache &c = g_eval_cache;
        if (c.device != device) {
            if (c.buf) (void)hipFree(c.buf);
            if (c.s) (void)hipStreamDestroy(c.s);
            c.buf = nullptr;
            c.s = nullptr;
            c.cap = 0;
            c.device = device;
        }
        if (!c.s && hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            c.s = nullptr;
            return nullptr;
        }
        if (c.cap < bytes) {
            if (c.buf) (void)hipFree(c.buf);
            c.buf = nullptr;
            c.cap = 0;
            if (hipMalloc(&c.buf, bytes) != hipSuccess) {
                (void)hipGetLastError();
                c.buf = nullptr;
                return nullptr;
            }
            c.cap = bytes;
        }
        *s_out = c.s;
        return c.buf;
    }
};
} // namespace
void sc_internal_release_eval_cache() { // sc_release_caches (gkr.hip)
    std::lock_guard<std::mutex> lk(g_eval_cache.mu);
    if (g_eval_cache.device >= 0) (void)hipSetDevice(g_eval_cache.device);
    if (g_eval_cache.buf) (void)hipFree(g_eval_cache.buf);
    if (g_eval_cache.s) (void)hipStreamDestroy(g_eval_cache.s);
    g_eval_cache.buf = nullptr;
    g_eval_cache.s = nullptr;
    g_eval_cache.cap = 0;
    g_eval_cache.device = -1;
}

extern "C" int sc_fix_variables(const uint64_t *in, uint32_t nv, const uint64_t *point, uint32_t k, uint64_t *out, uint32_t flags) {
    if (!in || !out || (k && !point)) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (k > nv || nv > 40) return sc_internal_fail(SC_ERR_BAD_ARG, "invalid partial point dimension"); // ark-poly's assert
    std::vector<sch::Fr> pt(k);
    for (uint32_t i = 0; i < k; ++i) {
        std::memcpy(&pt[i], point + 4 * i, 32);
        if (sch::geq_p(pt[i])) return sc_internal_fail(SC_ERR_BAD_ARG, "point[%u] is not a canonical field element", i);
    }
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    DeviceGate gate_(g_device);
    HIP_TRY(hipSetDevice(g_device));
    const bool on_device = flags & SC_TABLES_ON_DEVICE;
    const uint64_t n = 1ULL << nv;
    // The k variables are bound three per pass (k_fold_multi: 8 entries in, 1 out, the order of ark-poly's fix_variables), so the table
    // moves (1 + 1/8 + ...) x its size instead of once per variable.  Work areas: the outputs of passes 1 and 2 (ping-pong from there on),
    // plus a staging copy of a host table; leased from sc_poly_evaluate's cache when it is free.
    std::vector<int> levels;
    for (uint32_t left = k; left > 0;) {
        const int l = left >= 3 ? 3 : (int)left;
        levels.push_back(l);
        left -= l;
    }
    const uint64_t na = levels.empty() ? 1 : n >> levels[0];
    const uint64_t nb = levels.size() < 2 ? 1 : na >> levels[1];
    const size_t bytes_stage = on_device ? 0 : (size_t)n * 32, bytes_a = (size_t)na * 32, bytes_b = (size_t)nb * 32;
    EvalLease lease;
    DevMem own;
    StreamGuard sg;
    hipStream_t s = nullptr;
    char *base = static_cast<char *>(lease.get(g_device, bytes_stage + bytes_a + bytes_b, &s));
    if (!base) {
        HIP_TRY(hipMalloc(&own.p, bytes_stage + bytes_a + bytes_b));
        HIP_TRY(hipStreamCreateWithFlags(&sg.s, hipStreamNonBlocking));
        base = static_cast<char *>(own.p);
        s = sg.s;
    }
    uint4 *area[2] = {reinterpret_cast<uint4 *>(base + bytes_stage), reinterpret_cast<uint4 *>(base + bytes_stage + bytes_a)};
    const uint4 *cur = reinterpret_cast<const uint4 *>(in);
    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(base, in, n * 32, hipMemcpyHostToDevice, s));
        cur = reinterpret_cast<const uint4 *>(base);
    }
    uint64_t m = n;
    uint32_t var = 0;
    for (size_t ps = 0; ps < levels.size(); ++ps) {
        const int L = levels[ps];
        m >>= L;
        const bool last = ps + 1 == levels.size();
        scd::FoldArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.src[0] = cur;
        fa.dst[0] = (last && on_device) ? reinterpret_cast<uint4 *>(out) : area[ps & 1];
        for (int l = 0; l < L; ++l) {
            sch::Fr r32v = pt[var + l]; // r * 2^5 for the 2^261-radix arithmetic
            for (int dbl = 0; dbl < 5; ++dbl) r32v = sch::add(r32v, r32v);
            fa.r32[l] = to_dev(r32v);
        }
        HIP_TRY(scd::launch_fold_multi(fa, L, 1, m, s));
        cur = fa.dst[0];
        var += L;
    }
    if (!(k > 0 && on_device)) // (nothing bound: the table itself; host tables: the result comes back)
        HIP_TRY(hipMemcpyAsync(out, cur, m * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

extern "C" int sc_poly_evaluate(const sc_poly_desc *d, const uint64_t *point, uint64_t *out_value, uint64_t *out_table_values_or_null) {
    if (!d || !out_value || (d->num_vars && !point)) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (d->num_vars > 40) return sc_internal_fail(SC_ERR_BAD_ARG, "num_vars %u too large", d->num_vars);
    if (d->n_tables == 0 || !d->tables) return sc_internal_fail(SC_ERR_BAD_ARG, "no tables");
    if (d->n_products && (!d->coeffs || !d->prod_offsets || !d->prod_indices)) return sc_internal_fail(SC_ERR_BAD_ARG, "null product arrays");
    for (uint32_t k = 0; k < d->n_products; ++k) {
        if (d->prod_offsets[k + 1] <= d->prod_offsets[k]) return sc_internal_fail(SC_ERR_BAD_ARG, "product %u is empty", k);
        for (uint32_t q = d->prod_offsets[k]; q < d->prod_offsets[k + 1]; ++q)
            if (d->prod_indices[q] >= d->n_tables) return sc_internal_fail(SC_ERR_BAD_ARG, "product %u refers to table %u >= %u", k, d->prod_indices[q], d->n_tables);
    }
    for (uint32_t u = 0; u < d->n_tables; ++u)
        if (!d->tables[u]) return sc_internal_fail(SC_ERR_BAD_ARG, "table %u is null", u);
    const uint32_t nv = d->num_vars, U = d->n_tables;
    std::vector<sch::Fr> pt(nv);
    for (uint32_t i = 0; i < nv; ++i) {
        std::memcpy(&pt[i], point + 4 * i, 32);
        if (sch::geq_p(pt[i])) return sc_internal_fail(SC_ERR_BAD_ARG, "point[%u] is not a canonical field element", i);
    }
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    DeviceGate gate_(g_device);
    HIP_TRY(hipSetDevice(g_device));
    const bool on_device = d->flags & SC_TABLES_ON_DEVICE;
    const uint64_t n = 1ULL << nv;
    // passes: three variables at a time from the LSB end, the remainder (1 or 2) last, on a table that is tiny by then
    std::vector<int> levels;
    for (uint32_t left = nv; left > 0;) {
        const int l = left >= 3 ? 3 : (int)left;
        levels.push_back(l);
        left -= l;
    }
    // device memory: staging for host tables, two ping-pong work areas (sizes after pass 1 and pass 2), the U results
    const uint64_t na = levels.empty() ? 1 : n >> levels[0];
    const uint64_t nb = levels.size() < 2 ? 1 : na >> levels[1];
    const size_t bytes_stage = on_device ? 0 : (size_t)U * n * 32, bytes_a = (size_t)U * na * 32, bytes_b = (size_t)U * nb * 32, bytes_v = (size_t)U * 32;
    struct Area { // a view into the leased buffer, or an allocation of this call
        void *p = nullptr;
    } stage, wa, wb, vals;
    EvalLease lease;
    DevMem own;     // this call's allocation when the cache is taken or too small to grow
    StreamGuard sg; // ... and its stream
    hipStream_t s_eval = nullptr;
    char *base = static_cast<char *>(lease.get(g_device, bytes_stage + bytes_a + bytes_b + bytes_v, &s_eval));
    if (!base) {
        HIP_TRY(hipMalloc(&own.p, bytes_stage + bytes_a + bytes_b + bytes_v));
        HIP_TRY(hipStreamCreateWithFlags(&sg.s, hipStreamNonBlocking));
        base = static_cast<char *>(own.p);
        s_eval = sg.s;
    }
    stage.p = base;
    wa.p = base + bytes_stage;
    wb.p = base + bytes_stage + bytes_a;
    vals.p = base + bytes_stage + bytes_a + bytes_b;
    std::vector<const uint4 *> cur(U);
    for (uint32_t u = 0; u < U; ++u) {
        if (on_device) {
            cur[u] = reinterpret_cast<const uint4 *>(d->tables[u]);
        } else {
            uint4 *dst = static_cast<uint4 *>(stage.p) + 2 * n * u;
            HIP_TRY(hipMemcpyAsync(dst, d->tables[u], n * 32, hipMemcpyHostToDevice, s_eval));
            cur[u] = dst;
        }
    }
    uint64_t m = n;
    uint32_t var = 0;
    for (size_t ps = 0; ps < levels.size(); ++ps) {
        const int L = levels[ps];
        m >>= L;
        const bool last = ps + 1 == levels.size();
        uint4 *area = static_cast<uint4 *>((ps & 1) ? wb.p : wa.p);
        for (uint32_t u0 = 0; u0 < U; u0 += (uint32_t)scd::kMaxSmallTables) {
            const uint32_t cnt = std::min<uint32_t>(U - u0, (uint32_t)scd::kMaxSmallTables);
            scd::FoldArgs fa;
            std::memset(&fa, 0, sizeof(fa));
            for (uint32_t j = 0; j < cnt; ++j) {
                fa.src[j] = cur[u0 + j];
                fa.dst[j] = last ? static_cast<uint4 *>(vals.p) + 2 * (u0 + j) : area + 2 * m * (u0 + j);
            }
            for (int l = 0; l < L; ++l) {
                sch::Fr r32v = pt[var + l]; // r * 2^5 for the 2^261-radix arithmetic
                for (int dbl = 0; dbl < 5; ++dbl) r32v = sch::add(r32v, r32v);
                fa.r32[l] = to_dev(r32v);
            }
            HIP_TRY(scd::launch_fold_multi(fa, L, (int)cnt, m, s_eval));
            for (uint32_t j = 0; j < cnt; ++j) cur[u0 + j] = fa.dst[j];
        }
        var += L;
    }
    std::vector<sch::Fr> tv(U);
    if (levels.empty()) { // zero variables: a table is its single entry
        for (uint32_t u = 0; u < U; ++u) HIP_TRY(hipMemcpyAsync(&tv[u], cur[u], 32, hipMemcpyDeviceToHost, s_eval));
    } else {
        HIP_TRY(hipMemcpyAsync(tv.data(), vals.p, (size_t)U * 32, hipMemcpyDeviceToHost, s_eval));
    }
    HIP_TRY(hipStreamSynchronize(s_eval));
    sch::Fr acc = sch::zero();
    for (uint32_t k = 0; k < d->n_products; ++k) {
        sch::Fr pr;
        std::memcpy(&pr, d->coeffs + 4 * k, 32);
        if (sch::geq_p(pr)) return sc_internal_fail(SC_ERR_BAD_ARG, "coefficient %u is not a canonical field element", k);
        for (uint32_t q = d->prod_offsets[k]; q < d->prod_offsets[k + 1]; ++q) pr = sch::mul(pr, tv[d->prod_indices[q]]);
        acc = sch::add(acc, pr);
    }
    std::memcpy(out_value, &acc, 32);
    if (out_table_values_or_null) std::memcpy(out_table_values_or_null, tv.data(), (size_t)U * 32);
    return SC_OK;
}

// One-shot proofs (MLSumcheck::prove(&poly) in a loop, the reference's calling convention) and interactive provers (prover_init,
// prove_round x n, drop) would build and free a prover per use: device and pinned allocations, events, a stream, metadata uploads --
// 2 ms against a 0.4 ms proof at 2^16 entries.  The last prover that was freed is therefore kept (one, process-wide, arena at most
// kPoolMaxArena) and the next sc_prover_init / sc_ml_prove with the same polynomial STRUCTURE on the same device rewinds it onto the
// new tables (sc_prover_reset) instead.  sc_release_caches frees the kept one.
// (default of sc_set_cache_limit: 16 GiB of 288 GB; building and freeing a 4.5 GB arena costs 3 ms)
static std::atomic<uint64_t> g_cache_limit{16ULL << 30};
uint64_t sc_internal_cache_limit() { return g_cache_limit.load(std::memory_order_relaxed); } // gkr.hip
struct HandlePool {
    std::mutex mu;
    sc_prover *h = nullptr;
};
static HandlePool g_pool;
std::vector<uint8_t> pool_key_of(const sc_poly_desc *d, int device) {
    std::vector<uint8_t> k;
    auto put = [&](const void *p, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        k.insert(k.end(), b, b + n);
    };
    // (the policies a handle is BUILT under are part of its identity: a handle built before sc_set_policy is not the one to reuse after)
    const uint32_t head[8] = {d->num_vars, d->max_multiplicands, d->n_products, d->n_tables, d->flags, (uint32_t)device,
                              (uint32_t)scd::policy(scd::kPolWideTree), (uint32_t)scd::policy(scd::kPolTail)};
    put(head, sizeof(head));
    if (d->n_products) {
        put(d->prod_offsets, (size_t)(d->n_products + 1) * 4);
        put(d->prod_indices, (size_t)d->prod_offsets[d->n_products] * 4);
        put(d->coeffs, (size_t)d->n_products * 32);
    }
    return k;
}
sc_prover *handle_pool_take(const std::vector<uint8_t> &key) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (!g_pool.h || g_pool.h->pool_key != key) return nullptr;
    sc_prover *p = g_pool.h;
    g_pool.h = nullptr;
    return p;
}
bool handle_pool_offer(sc_prover *p) {
    if (p->pool_key.empty() || p->arena_bytes > sc_internal_cache_limit() || p->streamed) return false;
    if (p->stream != p->own_stream) return false; // (it runs on a stream of the caller's: sc_prover_set_stream)
    abandon_deferred(p);                          // nothing of it may still be waiting in the queue
    {   // sc_prover_free promises that the handle's work is over: asynchronous calls (sc_prove_round_partial, sc_prover_bind_final) may
        // still be reading borrowed tables or writing a caller's d_out, and after the free the caller has no stream left to wait on
        DeviceGate gate_(p->device);
        (void)hipSetDevice(p->device);
        (void)hipStreamSynchronize(p->own_stream);
    }
    if (p->timing) (void)sc_prover_set_timing(p, 0);
    sc_prover *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        old = g_pool.h;
        g_pool.h = p;
    }
    if (old) prover_destroy(old);
    return true;
}
void sc_internal_release_handle_pool() { // sc_release_caches (gkr.hip)
    sc_prover *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        old = g_pool.h;
        g_pool.h = nullptr;
    }
    if (old) prover_destroy(old);
}

// ---- library policy and launch-plan counters (kernels.h: PolicyKey, Plan) --------------------------------------------------------------
namespace {
struct PolicyDef {
    const char *name;
    int64_t def, lo, hi;
};
constexpr PolicyDef kPolicyDefs[scd::kPolCount] = {
    {"pipeline", 1, 0, 1},     {"resident", 1, 0, 1},          {"tail_slices", 1, 0, 1}, {"vram_mailbox", 1, 0, 1},           {"wide_tree", 1, 0, 1},
    {"rccl_direct", 1, 0, 1},  {"shard_gather_log2", 15, 1, 15}, {"gkr_direct", 1, 0, 1},  {"wait_spins", 1 << 22, 1, 0xffffffffLL}, {"tail", 1, 0, 1},
    {"staged_init", 1, 0, 1},
};
struct PolicyTable {
    std::atomic<int64_t> v[scd::kPolCount];
    PolicyTable() {
        for (int i = 0; i < scd::kPolCount; ++i) {
            int64_t x = kPolicyDefs[i].def;
#ifdef SC_EXPERIMENTS // the cross-check build (tests/test_gpu_variants.py) takes the initial values from SC_<KEY> in the environment
            std::string env = "SC_";
            for (const char *c = kPolicyDefs[i].name; *c; ++c) env += (char)std::toupper((unsigned char)*c);
            if (const char *e = std::getenv(env.c_str())) x = std::min(std::max<int64_t>(std::atoll(e), kPolicyDefs[i].lo), kPolicyDefs[i].hi);
#endif
            v[i].store(x, std::memory_order_relaxed);
        }
    }
};
PolicyTable &policy_table() {
    static PolicyTable t;
    return t;
}
std::atomic<uint64_t> g_plan[scd::kPlanCount];
constexpr const char *kPlanNames[scd::kPlanCount] = {
    "big.merged.round1", "big.merged.bind_chain", "big.merged.bind", "big.claim_identity", "big.store_f29", "big.store_canonical", "big.per_product_tree",
    "big.wide", "big.wide16", "big.generic", "big.node_by_node", "big.bind_pass", "big.streamed", "big.staged_round1", "finalize.multi_block", "finalize.one_block", "finalize.no_lds",
    "small.launched", "small.combos_table", "small.ptrs", "small.pipelined", "tail.slices8", "tail.slices12", "tail.rounds", "resident.slices",
    "resident.rounds", "sharded.rccl_direct", "sharded.rccl_publish", "sharded.host", "sharded.p2p", "sharded.gather_tail", "gkr.bucketed_grouped",
    "gkr.bucketed_counted", "gkr.list_form", "gkr.coeff_from_bound_table", "gkr.sharded", "fold_multi",
};
} // namespace
int64_t scd::policy(int key) { return key >= 0 && key < scd::kPolCount ? policy_table().v[key].load(std::memory_order_relaxed) : 0; }
void scd::plan_hit(int plan) {
    if (plan >= 0 && plan < scd::kPlanCount) g_plan[plan].fetch_add(1, std::memory_order_relaxed);
}
extern "C" int sc_set_policy(const char *key, int64_t value) {
    if (!key) return sc_internal_fail(SC_ERR_BAD_ARG, "null policy key");
    for (int i = 0; i < scd::kPolCount; ++i) {
        if (std::strcmp(key, kPolicyDefs[i].name) != 0) continue;
        if (value < kPolicyDefs[i].lo || value > kPolicyDefs[i].hi)
            return sc_internal_fail(SC_ERR_BAD_ARG, "policy %s takes %lld..%lld", key, (long long)kPolicyDefs[i].lo, (long long)kPolicyDefs[i].hi);
        policy_table().v[i].store(value, std::memory_order_relaxed);
        return SC_OK;
    }
    return sc_internal_fail(SC_ERR_BAD_ARG, "unknown policy key '%s'", key);
}
extern "C" int sc_get_policy(const char *key, int64_t *value) {
    if (!key || !value) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    for (int i = 0; i < scd::kPolCount; ++i) {
        if (std::strcmp(key, kPolicyDefs[i].name) != 0) continue;
        *value = scd::policy(i);
        return SC_OK;
    }
    return sc_internal_fail(SC_ERR_BAD_ARG, "unknown policy key '%s'", key);
}
extern "C" uint32_t sc_plan_count(void) { return (uint32_t)scd::kPlanCount; }
extern "C" const char *sc_plan_name(uint32_t i) { return i < (uint32_t)scd::kPlanCount ? kPlanNames[i] : nullptr; }
extern "C" int sc_plan_stats(uint64_t *out, uint32_t n) {
    if (!out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    for (uint32_t i = 0; i < n; ++i) out[i] = i < (uint32_t)scd::kPlanCount ? g_plan[i].load(std::memory_order_relaxed) : 0;
    return SC_OK;
}

extern "C" int sc_library_stats(uint64_t *out, uint32_t n) {
    if (!out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    for (uint32_t i = 0; i < n; ++i) out[i] = i < 8 ? g_stat[i].load(std::memory_order_relaxed) : 0;
    return SC_OK;
}

extern "C" int sc_set_cache_limit(uint64_t bytes) {
    const uint64_t before = g_cache_limit.exchange(bytes);
    return bytes < before ? sc_release_caches() : SC_OK;
}

extern "C" int sc_prover_set_resident(sc_prover *p, uint32_t patience_polls) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    p->resident_spins = patience_polls;
    return SC_OK;
}

extern "C" int sc_prover_set_polling(sc_prover *p, int allow) {
    if (!p) return sc_internal_fail(SC_ERR_BAD_ARG, "null prover");
    if (p->deferred_pending) return sc_internal_fail(SC_ERR_BAD_ARG, "a pipelined round is waiting for its challenge");
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    p->pipeline_ok = allow != 0;
    p->polling_off_by_caller = allow == 0;
    return SC_OK;
}


// ---------------------------------------------------------------------------------------------------
// verifier side (host; O(nv * deg) scalar work) -- reference src/ml_sumcheck/protocol/verifier.rs
// ---------------------------------------------------------------------------------------------------
sch::Fr fr_from_u128(sch::u128 x) { // F::from(u128): lo + hi * 2^64
    const sch::Fr lo = sch::from_u64((uint64_t)x), hi = sch::from_u64((uint64_t)(x >> 64));
    static const sch::Fr two64 = sch::mul(sch::Fr{{0, 1, 0, 0}}, sch::kR2);
    return sch::add(lo, sch::mul(hi, two64));
}

// verifier.rs:139-251, literally: the value at x of the unique polynomial of degree < len through (i, y[i]), with the same
// three tiers for the ratio of denominators (machine i64 / i128 / field) and the same early returns.  Every tier computes
// the same field value; they are kept so that a reader can diff this against the reference line by line.
sch::Fr interpolate(const sch::Fr *y, uint32_t len, const sch::Fr &x) {
    std::vector<sch::Fr> evals;
    evals.reserve(len);
    sch::Fr prod = x;
    evals.push_back(x);
    sch::Fr check = sch::zero();
    for (uint32_t i = 1; i < len; ++i) { // verifier.rs:150-160
        if (sch::eq(x, check)) return y[i - 1];
        check = sch::add(check, sch::kOne);
        const sch::Fr tmp = sch::sub(x, check);
        evals.push_back(tmp);
        prod = sch::mul(prod, tmp);
    }
    if (sch::eq(x, check)) return y[len - 1]; // verifier.rs:162-164
    sch::Fr res = sch::zero();
    auto term = [&](uint32_t i, const sch::Fr &num, const sch::Fr &den) { // res += p_i[i] * prod * num / (den * evals[i])
        res = sch::add(res, sch::mul(sch::mul(sch::mul(y[i], prod), num), sch::inverse(sch::mul(den, evals[i]))));
    };
    if (len <= 20) { // verifier.rs:193-213: i64 / u64 ratio
        uint64_t fact = 1;
        for (uint32_t k = 2; k < len; ++k) fact *= k;
        const sch::Fr last_denom = sch::from_u64(fact);
        int64_t ratio_numerator = 1;
        uint64_t ratio_enumerator = 1;
        for (uint32_t i = len; i-- > 0;) {
            const sch::Fr rn = ratio_numerator < 0 ? sch::neg(sch::from_u64((uint64_t)(-ratio_numerator))) : sch::from_u64((uint64_t)ratio_numerator);
            term(i, sch::from_u64(ratio_enumerator), sch::mul(last_denom, rn));
            if (i != 0) {
                ratio_numerator *= -((int64_t)len - (int64_t)i);
                ratio_enumerator *= (uint64_t)i;
            }
        }
    } else if (len <= 33) { // verifier.rs:214-234: i128 / u128 ratio
        sch::u128 fact = 1;
        for (uint32_t k = 2; k < len; ++k) fact *= k;
        const sch::Fr last_denom = fr_from_u128(fact);
        __int128 ratio_numerator = 1;
        sch::u128 ratio_enumerator = 1;
        for (uint32_t i = len; i-- > 0;) {
            const sch::Fr rn = ratio_numerator < 0 ? sch::neg(fr_from_u128((sch::u128)(-ratio_numerator))) : fr_from_u128((sch::u128)ratio_numerator);
            term(i, fr_from_u128(ratio_enumerator), sch::mul(last_denom, rn));
            if (i != 0) {
                ratio_numerator *= -((__int128)len - (__int128)i);
                ratio_enumerator *= (sch::u128)i;
            }
        }
    } else { // verifier.rs:235-248: the ratio as field elements
        sch::Fr denom_up = sch::kOne; // field_factorial(len - 1)
        for (uint32_t k = 1; k < len; ++k) denom_up = sch::mul(denom_up, sch::from_u64(k));
        sch::Fr denom_down = sch::kOne;
        for (uint32_t i = len; i-- > 0;) {
            term(i, denom_down, denom_up);
            if (i != 0) {
                denom_up = sch::mul(denom_up, sch::neg(sch::from_u64(len - i)));
                denom_down = sch::mul(denom_down, sch::from_u64(i));
            }
        }
    }
    return res;
}

// every element handed to the verifier must be a canonical Montgomery residue (< p): the reference's Fp cannot hold anything
// else, and host_fr.hpp's add() assumes it (a non-canonical ev0 + p, ev1 + p would wrap past 2^256 and pass the sum check)
int require_canonical(const uint64_t *limbs, size_t n_elems, const char *what) {
    for (size_t i = 0; i < n_elems; ++i) {
        sch::Fr v;
        std::memcpy(&v, limbs + 4 * i, 32);
        if (sch::geq_p(v)) return sc_internal_fail(SC_ERR_BAD_ARG, "%s element %zu is not a canonical field element", what, i);
    }
    return SC_OK;
}

extern "C" int sc_interpolate_uni_poly(const uint64_t *p_i, uint32_t len, const uint64_t *eval_at, uint64_t *out) {
    if (!p_i || !eval_at || !out || len == 0) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    int rc = require_canonical(p_i, len, "p_i");
    if (!rc) rc = require_canonical(eval_at, 1, "eval_at");
    if (rc) return rc;
    sch::Fr x;
    std::memcpy(&x, eval_at, 32);
    const sch::Fr v = interpolate(reinterpret_cast<const sch::Fr *>(p_i), len, x);
    std::memcpy(out, v.l, 32);
    return SC_OK;
}

extern "C" int sc_ml_verify(uint32_t num_vars, uint32_t max_multiplicands, const uint64_t *claimed_sum, const uint64_t *proof,
                            uint64_t proof_elems, sc_rng *rng_or_null, uint64_t *out_point, uint64_t *out_expected) {
    if (!claimed_sum || !proof || !out_point || !out_expected) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    // every message is read at [0] and [1] (verifier.rs:101-102; a one-element message makes the reference panic on the index): a
    // polynomial without multiplicands has no valid proof, and max_multiplicands + 1 must not wrap
    if (max_multiplicands == 0 || max_multiplicands == UINT32_MAX)
        return sc_internal_fail(SC_ERR_BAD_ARG, "max_multiplicands %u: a round message needs at least two evaluations", max_multiplicands);
    const uint32_t D = max_multiplicands + 1;
    // verifier.rs:60-62 panics on a message of the wrong length; here the caller states how many elements `proof` holds
    if (proof_elems != (uint64_t)num_vars * D) return sc_internal_fail(SC_ERR_BAD_ARG, "incorrect number of evaluations");
    int rc = require_canonical(claimed_sum, 1, "claimed_sum");
    if (!rc) rc = require_canonical(proof, (size_t)proof_elems, "proof");
    if (rc) return rc;
    sc_rng local;
    sch::Blake2b512Rng &rng = rng_or_null ? rng_or_null->rng : local.rng;
    rng.feed_poly_info(max_multiplicands, num_vars); // mod.rs:90
    const sch::Fr *msgs = reinterpret_cast<const sch::Fr *>(proof);
    std::vector<sch::Fr> rs(num_vars);
    for (uint32_t i = 0; i < num_vars; ++i) { // mod.rs:92-97, verify_round = store + sample (verifier.rs:54-83)
        rng.feed_prover_msg(msgs + (size_t)i * D, D);
        rs[i] = rng.sample_fr();
    }
    sch::Fr expected;
    std::memcpy(&expected, claimed_sum, 32);
    for (uint32_t i = 0; i < num_vars; ++i) { // check_and_generate_subclaim, verifier.rs:90-121
        const sch::Fr *ev = msgs + (size_t)i * D;
        if (!sch::eq(sch::add(ev[0], ev[1]), expected)) return sc_internal_fail(SC_ERR_REJECT, "Prover message is not consistent with the claim.");
        expected = interpolate(ev, D, rs[i]);
    }
    if (num_vars) std::memcpy(out_point, rs.data(), (size_t)num_vars * 32);
    std::memcpy(out_expected, expected.l, 32);
    return SC_OK;
}

// ---------------------------------------------------------------------------------------------------
// integer all-reduce lanes -> field
// ---------------------------------------------------------------------------------------------------
extern "C" int sc_wide_reduce(const uint64_t *wide, uint32_t n_elems, uint64_t *out) {
    if (!wide || !out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    for (uint32_t e = 0; e < n_elems; ++e) {
        // V = sum_j lane_j * 2^(32 j), lanes < 2^64  =>  V < 2^(64+224) ; keep 5 x u64
        uint64_t v[6] = {0, 0, 0, 0, 0, 0};
        for (int j = 0; j < 8; ++j) {
            const uint64_t lane = wide[8 * (size_t)e + j];
            const int w = j >> 1, sh = (j & 1) * 32;
            sch::u128 add = (sch::u128)lane << sh; // up to 96 bits
            sch::u128 c = (sch::u128)v[w] + (uint64_t)add;
            v[w] = (uint64_t)c;
            c = (c >> 64) + (uint64_t)(add >> 64);
            for (int q = w + 1; q < 6 && c != 0; ++q) {
                c += v[q];
                v[q] = (uint64_t)c;
                c >>= 64;
            }
        }
        if (v[5] != 0) return sc_internal_fail(SC_ERR_BAD_ARG, "wide lanes overflow");
        // V = lo + hi * 2^256, hi < 2^64:  V mod p = (lo mod p) + hi * R   where R = 2^256 mod p = mont_mul(hi, R^2)
        sch::Fr lo = {{v[0], v[1], v[2], v[3]}};
        while (sch::geq_p(lo)) lo = sch::sub_p(lo); // lo < 2^256 < 3p: at most two subtractions
        const sch::Fr hi = sch::mul(sch::Fr{{v[4], 0, 0, 0}}, sch::kR2);
        const sch::Fr res = sch::add(lo, hi);
        std::memcpy(out + 4 * (size_t)e, res.l, 32);
    }
    return SC_OK;
}

// ---------------------------------------------------------------------------------------------------
// synthetic inputs + instrumentation
// ---------------------------------------------------------------------------------------------------
extern "C" int sc_synth_table_device(uint64_t seed, uint64_t stream, uint64_t first, uint64_t n, uint64_t *d_out) {
    if (!d_out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible");
    HIP_TRY(hipSetDevice(g_device));
    HIP_TRY(scd::launch_synth(seed, stream, first, n, reinterpret_cast<uint4 *>(d_out), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return SC_OK;
}

extern "C" int sc_claim_weights(uint32_t M, const uint64_t *r, uint64_t *out) {
    if (!r || !out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (M < 1 || M > 4) return sc_internal_fail(SC_ERR_BAD_ARG, "claim weights exist for 1..4 multiplicands, not %u", M);
    sch::Fr rr, lam[5];
    std::memcpy(&rr, r, 32);
    if (sch::geq_p(rr)) return sc_internal_fail(SC_ERR_BAD_ARG, "the point is not a canonical field element");
    claim_weights(M, rr, lam);
    std::memcpy(out, lam, (size_t)(M + 1) * 32);
    return SC_OK;
}

extern "C" int sc_fr_elementwise(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n) {
    if (!a || !b || !out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (op < 0 || op > 5) return sc_internal_fail(SC_ERR_BAD_ARG, "unknown op %d", op);
    if (n == 0) return SC_OK;
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    HIP_TRY(hipSetDevice(g_device));
    void *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&da, n * 32));
    HIP_TRY(hipMalloc(&db, n * 32));
    HIP_TRY(hipMalloc(&dout, n * 32));
    HIP_TRY(hipMemcpy(da, a, n * 32, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, n * 32, hipMemcpyHostToDevice));
    FrHost u;
    std::memcpy(&u, b, 32); // op 5 multiplies every a[i] by the uniform element b[0]
    HIP_TRY(scd::launch_fr_elementwise(op, static_cast<const uint4 *>(da), static_cast<const uint4 *>(db), u, static_cast<uint4 *>(dout), n, nullptr));
    HIP_TRY(hipMemcpy(out, dout, n * 32, hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return SC_OK;
}

extern "C" int sc_bench_modmul(uint64_t n_threads, uint32_t reps, uint32_t variant, float *ms_out, uint64_t *checksum_out) {
    if (!ms_out) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible");
    HIP_TRY(hipSetDevice(g_device));
    uint64_t *d_sink = nullptr;
    hipEvent_t e0, e1;
    HIP_TRY(hipMalloc(&d_sink, 8));
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(scd::launch_bench_modmul(n_threads, 8, variant, d_sink, nullptr)); // warm-up
    HIP_TRY(hipEventRecord(e0, nullptr));
    HIP_TRY(scd::launch_bench_modmul(n_threads, reps, variant, d_sink, nullptr));
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms_out, e0, e1));
    if (checksum_out) HIP_TRY(hipMemcpy(checksum_out, d_sink, 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_sink);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return SC_OK;
}

