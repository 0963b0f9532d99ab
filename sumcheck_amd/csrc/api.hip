// api.hip -- the C ABI of libsumcheck_hip.so (declared in include/sumcheck_hip.h): prover state
// resident in HBM, the per-round launch plan, and the host-side protocol drivers that the reference
// runs around prove_round (reference src/ml_sumcheck/mod.rs:50-70).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums only: the entry points are bound with dlsym

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sumcheck_hip.h"
#include "host_fr.hpp"
#include "kernels.h"
#include "transcript.hpp"

using scd::FinProd;
using scd::FrHost;
using scd::ProdArgs;
using scd::Combo;
using scd::TablePtrs;

// ---------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------
// How long a host loop waits for a round's message (or a peer's lanes) before it declares the proof dead: sc_set_publish_timeout_ms,
// SC_PUBLISH_TIMEOUT_MS in the environment, 20 s by default (a device-side wait's own bound -- SC_WAIT_SPINS -- expires long before).
static std::atomic<uint32_t> g_publish_timeout_ms{0}; // 0: not set yet
static std::chrono::milliseconds publish_timeout() {
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
static thread_local uint16_t g_gate_depth[64];
static void gate_lock(int device) {
    g_gate_mutex[(unsigned)device & 63u].lock();
    ++g_gate_depth[(unsigned)device & 63u];
}
static void gate_unlock(int device) {
    --g_gate_depth[(unsigned)device & 63u];
    g_gate_mutex[(unsigned)device & 63u].unlock();
}
struct DeviceGate {
    const int device;
    explicit DeviceGate(int d) : device(d) { gate_lock(device); }
    ~DeviceGate() { gate_unlock(device); }
    DeviceGate(const DeviceGate &) = delete;
    DeviceGate &operator=(const DeviceGate &) = delete;
};
struct GateYield { // drop every level this thread holds, take them back on scope exit
    const int device;
    uint16_t depth;
    GateYield(int d, bool enable) : device(d), depth(enable ? g_gate_depth[(unsigned)d & 63u] : 0) {
        for (uint16_t i = 0; i < depth; ++i) gate_unlock(device);
    }
    ~GateYield() {
        for (uint16_t i = 0; i < depth; ++i) gate_lock(device);
    }
    GateYield(const GateYield &) = delete;
    GateYield &operator=(const GateYield &) = delete;
};
void sc_internal_gate_lock(int device) { gate_lock(device); } // gkr.hip
void sc_internal_gate_unlock(int device) { gate_unlock(device); }

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
uint64_t sc_internal_cache_limit(); // sc_set_cache_limit: what each process-wide cache may keep (defined with the handle pool)
// shared with gkr.hip
int sc_internal_device() { return g_device; } // the calling thread's device (sc_set_device), for gkr.hip

int sc_internal_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
// (a failed HIP call also leaves its code as the thread's sticky "last error": it is taken out here, or the next kernel launch of this
// thread -- whose wrapper returns hipGetLastError() -- would report it again: one refused hipMalloc must not fail the proofs after it)
#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess) {                                                                                         \
            (void)hipGetLastError();                                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? SC_ERR_OOM : SC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                                     \
        }                                                                                                               \
    } while (0)

static inline FrHost to_dev(const sch::Fr &a) {
    FrHost h;
    std::memcpy(&h, &a, sizeof(h));
    return h;
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
struct sc_rng {
    sch::Blake2b512Rng rng;
};
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
// ---------------------------------------------------------------------------------------------------
struct Product {
    sch::Fr coeff;
    std::vector<uint32_t> tables; // distinct tables of the product, first-occurrence order
    std::vector<uint32_t> exps;   // multiplicity of each
    uint32_t M = 0;               // number of multiplicands
    bool fused = false;           // M <= kMaxFusedM: register-resident kernel, bind fused in
    uint64_t partial_off = 0;     // element offset into d_partials
    uint32_t slot_off = 0;        // generic path: offset into d_slot_table / d_slot_exp
};

struct Table {
    const uint4 *cur = nullptr;       // this round's evaluations (main array)
    const int32_t *cur_top = nullptr; // non-null: `cur` is in the internal F29 format and this is its limb-8 array
    uint4 *buf[2] = {nullptr, nullptr};
    int32_t *buf_top[2] = {nullptr, nullptr}; // limb-8 arrays of the two ping-pong buffers
    int next = 0;                     // buffer the next bind writes to
};

constexpr uint32_t kResidentSpinsDefault = 256; // ~0.5 ms of polls
struct sc_prover {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    uint32_t nv = 0, max_mult = 0, D = 0, K = 0, U = 0, round = 0;
    bool exhausted = false;
    std::vector<sch::Fr> randomness;
    std::vector<Product> prods;
    std::vector<Table> tabs;
    void *arena = nullptr;
    FrHost *d_partials = nullptr;
    FrHost *d_partials2 = nullptr;    // in-kernel finalize of the merged big-round launch: per-group partial sums ...
    uint32_t *d_fin_mb_counter = nullptr; // (inside d_fin_counters)
    uint32_t *d_fin_counters = nullptr; // ... and its arrival counters (the kernel leaves them at zero)
    FinProd *d_finprods = nullptr;
    FrHost *d_W = nullptr; // node -> message matrices of every product (see FinProd::w_off)
    FrHost *d_scratch = nullptr;
    // the multi-block finalize's node sums of the last two rounds (K * D each, round & 1 selects): a big binding round whose predecessor's
    // sums are here leaves node 1 to the claim identity (kernels.h: ClaimArgs).  sums_round: the round whose complete sums are held, or -1
    FrHost *d_sums[2] = {nullptr, nullptr};
    int64_t sums_round = -1;
    FrHost *d_out = nullptr;
    FrHost *h_out = nullptr;      // pinned, host-mapped: k_finalize writes the message here directly
    uint32_t *h_flag = nullptr;   // pinned, host-mapped sequence flag raised by k_finalize
    FrHost *h_out_dev = nullptr;  // device-side aliases of the two
    uint32_t *h_flag_dev = nullptr;
    uint32_t seq = 0;
    uint64_t *d_wide = nullptr;      // sharded rounds inside the library: all-reduce buffer (D x 8 lanes) ...
    uint64_t *h_wide = nullptr;      // ... and its host-mapped landing page
    uint64_t *h_wide_dev = nullptr;
    void *d_tail_send = nullptr, *d_tail_recv = nullptr, *d_tail_tabs = nullptr; // sc_ml_prove_sharded: bind_final out, all-gather out, G-entry tables
    sc_prover *tail = nullptr;       // ... and the prover of the replicated last rounds over them (built once, rewound per proof)
    uint32_t tail_ranks = 0;
    size_t tail_buf_bytes = 0;       // size of d_tail_recv / d_tail_tabs as allocated (d_tail_send: a G-th of it)
    std::vector<std::vector<uint32_t>> prod_indices; // the descriptor's product lists as given (for the tail's descriptor)
    Combo *d_combos = nullptr;    // (product, point) combinations for the small-round kernel
    std::vector<FinProd> h_finprods; // host copy of d_finprods (kernel-argument path of k_finalize)
    scd::ComboMeta meta;          // the same metadata as a kernel argument (when it fits: has_meta)
    bool has_meta = false;
    int n_combos = 0;
    bool any_generic = false;
    const uint4 **d_cur_tables = nullptr;
    const uint4 **h_cur_tables = nullptr; // pinned
    uint32_t *d_slot_table = nullptr, *d_slot_exp = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // pipelined late rounds (sc_ml_prove_handle, GKR): the next round is enqueued behind a one-lane wait kernel before the
    // current round's message has been hashed; its bind kernel reads the challenge from the host-mapped mailbox
    uint32_t *sig = nullptr;        // host-mapped word the wait kernel polls (after the two mailbox slots)
    uint32_t *sig_dev = nullptr;
    uint32_t sig_seq = 0;           // last value waited for
    FrHost *h_mail = nullptr;       // host-mapped, two slots (+ the word above)
    FrHost *h_mail_dev = nullptr;
    FrHost *d_mail = nullptr;       // device-memory copy of the slot in use (two slots), filled by the wait kernel
    uint32_t *d_tail_sync = nullptr; // persistent tail kernel: 4 sync words + 2 challenge slots (device)
    int tail_max_blocks = 0;        // blocks of it the device holds at once (its grid never exceeds that)
    uint64_t arena_bytes = 0;       // size of the bound-table arena (what a pooled handle keeps allocated)
    std::vector<uint8_t> pool_key;  // non-empty: created by sc_ml_prove; sc_prover_free offers it back to the pool (handle_pool_*)
    uint32_t n_retries = 0;         // proofs repeated after an expired device-side wait (sc_ml_prove_handle)
    bool pipeline_ok = true;        // cleared when the wait-value path is unavailable (or SC_PIPELINE=0, SC_NO_DEVICE_POLLING, sc_prover_set_polling(p, 0))
    bool polling_off_by_caller = false; // ... by the caller: survives what re-enables pipeline_ok internally
    // the interactive sc_prove_round's resident kernel (k_tail_rounds kept across calls: see resident_start)
    struct Resident {
        bool active = false;
        bool first_has_bind = false;
        uint32_t seq0 = 0, sig0 = 0, n_rounds = 0, done = 0; // done: rounds whose message the host has taken
    } res;
    uint32_t resident_spins = kResidentSpinsDefault; // its patience for the next call, in polls of the host-mapped mailbox (~2 us each); 0: not used
    bool deferred_pending = false;  // a round is enqueued behind the wait and still needs its challenge
    bool fused_finalize = false;    // experiments, SC_FUSED_FIN=1: the merged big-round launch finalizes in-kernel (measured: slower than the k_finalize launch)
    bool use_tail = true;           // sc_ml_prove* / GKR: the latency-bound rounds run in the persistent tail kernel (SC_TAIL=0: pipelined launches)
    bool merge_rounds = false; // big rounds run as ONE launch over all products (k_round_tree): <= kMaxRoundProds products of <= 4 multiplicands
    bool use_f29 = false; // bound tables of big rounds kept in the internal 9 x 29-bit format (all products <= 4 multiplicands)
    // The production path is fixed: product tree, carry-free arithmetic.  A -DSC_EXPERIMENTS build (libsumcheck_hip_exp.so, used by
    // tests/test_gpu_variants.py) lets the environment select the cross-check kernels instead.
    bool use_fe = true;     // experiments: SC_FE=0 selects the saturated (Comba asm) kernels
    int kernel_variant = 3; // experiments: SC_KERNEL 0 = node by node (k_prod_round[_fe]), 2 = tiled LDS-staged (k_round_tile), 3 = product tree
    // streamed tables (SC_TABLES_STREAM): the inputs stay in HOST memory; rounds 1 and 2 pull them through a two-slot staging ring in
    // chunks, so HBM only ever holds the bound tables (from round 2 on everything is resident and the ordinary path continues)
    bool streamed = false;
    uint32_t stream_chunk_request = 0;    // sc_prover_init_streamed's chunk_log2 (0: default)
    uint32_t chunk_log2 = 0;              // entries of every table per chunk
    std::vector<const uint64_t *> host_tabs;
    void *ring[2] = {nullptr, nullptr};   // U x 2^chunk_log2 x 32 bytes each
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    FrHost *d_chunk_msg = nullptr;        // a chunk's message, and the running sum over the chunks (2 x D elements)
    // reset support + per-product instrumentation
    bool borrow = false;
    std::vector<const uint4 *> origin; // borrowed table pointers (borrow mode)
    bool timing = false, timing_pending = false;
    bool prod_merged = false; // ... as one event pair around the merged launch (attributed to product 0)
    bool prod_timed = false; // the pending round recorded per-product events (big rounds only)
    std::vector<hipEvent_t> prod_ev;   // 2 per product
    std::vector<double> prod_ms;       // accumulated device time of each product's kernel
    std::vector<uint64_t> prod_launches;
    double rounds_ms = 0.0;            // accumulated ev0..ev1 (all kernels of a round incl. finalize)
    std::vector<double> round_kernel_ms;   // per round (index = round - 1): accumulated device time of the merged big-round launch ...
    std::vector<uint64_t> round_kernel_launches; // ... and how many launches that is (sc_prover_get_round_timing)
    uint32_t timed_round = 0;          // the round the pending event pairs belong to
};

static int resident_quiesce(sc_prover *p); // the interactive protocol's resident kernel leaves before anything else touches the handle
static void prover_destroy(sc_prover *p) {
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

static bool handle_pool_offer(sc_prover *p);
extern "C" void sc_prover_free(sc_prover *p) {
    if (p) (void)resident_quiesce(p);
    if (p && handle_pool_offer(p)) return; // (a handle sc_ml_prove built: kept for the next proof of the same shape)
    prover_destroy(p);
}

static int validate_desc(const sc_poly_desc *d) {
    if (!d) return fail(SC_ERR_BAD_ARG, "null descriptor");
    if (d->num_vars == 0) return fail(SC_ERR_CONSTANT_POLY, "Attempt to prove a constant.");
    if (d->num_vars > 40) return fail(SC_ERR_BAD_ARG, "num_vars %u too large", d->num_vars);
    if (d->n_tables == 0 || !d->tables) return fail(SC_ERR_BAD_ARG, "no tables");
    if (d->n_products && (!d->coeffs || !d->prod_offsets || !d->prod_indices)) return fail(SC_ERR_BAD_ARG, "null product arrays");
    uint32_t mx = 0;
    for (uint32_t k = 0; k < d->n_products; ++k) {
        if (d->prod_offsets[k + 1] <= d->prod_offsets[k]) return fail(SC_ERR_BAD_ARG, "product %u is empty", k); // data_structures.rs:78
        mx = std::max(mx, d->prod_offsets[k + 1] - d->prod_offsets[k]);
        for (uint32_t q = d->prod_offsets[k]; q < d->prod_offsets[k + 1]; ++q)
            if (d->prod_indices[q] >= d->n_tables) return fail(SC_ERR_BAD_ARG, "product %u refers to table %u >= %u", k, d->prod_indices[q], d->n_tables);
    }
    if (mx != d->max_multiplicands) return fail(SC_ERR_BAD_ARG, "max_multiplicands %u != max product length %u", d->max_multiplicands, mx);
    for (uint32_t u = 0; u < d->n_tables; ++u)
        if (!d->tables[u]) return fail(SC_ERR_BAD_ARG, "table %u is null", u);
    return SC_OK;
}

static sch::Fr fr_small(int64_t v) { return v >= 0 ? sch::from_u64((uint64_t)v) : sch::neg(sch::from_u64((uint64_t)(-v))); }

// (deg+1) x (M+1) matrix taking a degree-M polynomial's values at the kernel nodes (scd::node_value: 0, 1, inf, -1, 2, ...)
// to its values at 0..deg, times `scale`.  Exact Lagrange weights in the field; "inf" is the leading coefficient L:
// P(t) = L t^M + sum_i (P(x_i) - L x_i^M) l_i(t) over the M finite nodes.
static void build_node_matrix(uint32_t M, uint32_t D, const sch::Fr &scale, std::vector<sch::Fr> &out) {
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
static void claim_weights(uint32_t M, const sch::Fr &r, sch::Fr *lam) {
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

static int prover_build(const sc_poly_desc *d, sc_prover *p) {
    DeviceGate gate_(g_device);
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
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
#ifdef SC_EXPERIMENTS
    if (const char *e = std::getenv("SC_F29")) p->use_f29 = p->use_f29 && std::atoi(e) != 0;
#endif
    for (uint32_t k = 0; k < d->n_products; ++k)
        if (d->prod_offsets[k + 1] - d->prod_offsets[k] > 4) p->use_f29 = false;
    // with more tables than the small-round kernels take, the big-round kernels also run the short rounds, whose tables are
    // smaller than one 128-entry block of the chunk-planar layout
    if (d->n_tables > (uint32_t)scd::kMaxSmallTables) p->use_f29 = false;
    p->merge_rounds = p->kernel_variant == 3 && d->n_products > 0 && d->n_products <= (uint32_t)scd::kMaxRoundProds;
    for (uint32_t k = 0; k < d->n_products; ++k)
        if (d->prod_offsets[k + 1] - d->prod_offsets[k] > 4) p->merge_rounds = false;
#ifdef SC_EXPERIMENTS
    if (const char *e = std::getenv("SC_MERGE")) p->merge_rounds = p->merge_rounds && std::atoi(e) != 0;
    if (const char *e = std::getenv("SC_FUSED_FIN")) p->fused_finalize = std::atoi(e) != 0;
    if (const char *e = std::getenv("SC_TAIL")) p->use_tail = std::atoi(e) != 0; // 0: late rounds as pipelined launches (the path sharded RCCL proofs take)
#endif

    // products: distinct tables + multiplicities
    uint64_t partial_elems = 0;
    std::vector<uint32_t> slot_table, slot_exp;
    std::vector<FinProd> fin(p->K);
    std::vector<Combo> combos;
    std::vector<sch::Fr> Wall;
    for (uint32_t k = 0; k < p->K; ++k) {
        Product pr;
        std::memcpy(&pr.coeff, d->coeffs + 4 * k, 32);
        if (sch::geq_p(pr.coeff)) return fail(SC_ERR_BAD_ARG, "coefficient %u is not a canonical field element", k);
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
    HIP_TRY(hipMalloc(&p->d_W, std::max<size_t>(Wall.size(), 1) * 32));
    if (!Wall.empty()) HIP_TRY(hipMemcpyAsync(p->d_W, Wall.data(), Wall.size() * 32, hipMemcpyHostToDevice, p->stream));
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
    if (p->any_generic) { // (two sets: a streamed handle's chunks alternate between them, as between the staging slots)
        HIP_TRY(hipMalloc(&p->d_cur_tables, 2 * p->U * sizeof(void *)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_cur_tables), 2 * p->U * sizeof(void *), hipHostMallocDefault));
    }
    HIP_TRY(hipStreamSynchronize(p->stream)); // inputs are copied: the caller may drop them now (prover.rs:55-59)
    return SC_OK;
}

static std::vector<uint8_t> pool_key_of(const sc_poly_desc *d, int device);
static sc_prover *handle_pool_take(const std::vector<uint8_t> &key);
// Per-owner policy of a handle, as prover_build leaves it: whether kernels may wait for the host (SC_NO_DEVICE_POLLING /
// sc_prover_set_polling), the resident kernel's patience (sc_prover_set_resident), per-launch timing (sc_prover_set_timing).  A handle
// that comes back from the pool starts from these, whatever its previous owner had set.
static void reset_owner_policy(sc_prover *p, uint32_t desc_flags) {
    p->polling_off_by_caller = (desc_flags & SC_NO_DEVICE_POLLING) != 0;
    p->pipeline_ok = !p->polling_off_by_caller;
    p->resident_spins = kResidentSpinsDefault;
    if (p->timing) (void)sc_prover_set_timing(p, 0);
    p->n_retries = 0;
}
extern "C" int sc_prover_init(const sc_poly_desc *desc, sc_prover **out) {
    if (!out) return fail(SC_ERR_BAD_ARG, "null out");
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
    if (!p) return fail(SC_ERR_OOM, "host allocation failed");
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
    if (!out) return fail(SC_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (desc->flags & SC_TABLES_ON_DEVICE) return fail(SC_ERR_BAD_ARG, "streamed tables are host tables");
    sc_prover *p = new (std::nothrow) sc_prover();
    if (!p) return fail(SC_ERR_OOM, "host allocation failed");
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
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    DeviceGate gate_(p->device);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->stream = use_own ? p->own_stream : static_cast<hipStream_t>(hip_stream); // NULL = the legacy default stream
    return SC_OK;
}

// fold the previous round's event pairs into the accumulators (blocks until that round has finished)
static int collect_timing(sc_prover *p) {
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

// Launch one round's kernels on p->stream.  On return the round polynomial is in p->d_out (and in
// d_wide if non-null); nothing has been synchronised.
static uint64_t small_pairs_limit() { // the big/small round boundary
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
static bool launches_are_async(sc_prover *p) {
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
// tail kernel.  First use sets it up; any failure -- or a runtime that serialises launches, or SC_PIPELINE=0 -- switches both off
// for this handle.
static bool ensure_mailbox(sc_prover *p) {
    if (!p->pipeline_ok) return false;
    if (p->sig) return true;
    const char *env = std::getenv("SC_PIPELINE"); // read per handle, at its first late round
    bool env_off = env && std::atoi(env) == 0;
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
static bool skip1_enabled() {
#ifdef SC_NO_SKIP1
    return false;
#elif defined(SC_EXPERIMENTS)
    static const bool on = !(std::getenv("SC_SKIP1") && std::atoi(std::getenv("SC_SKIP1")) == 0);
    return on;
#else
    return true;
#endif
}
static bool fin_mb_enabled() {
#ifdef SC_EXPERIMENTS
    static const bool on = !(std::getenv("SC_FIN_MB") && std::atoi(std::getenv("SC_FIN_MB")) == 0);
    return on;
#else
    return true;
#endif
}
static bool can_defer_next(sc_prover *p) {
    if (!p->pipeline_ok || p->exhausted || p->round == 0 || p->round >= p->nv) return false;
    if (p->streamed && p->round < 2) return false; // round 2 of a streamed handle walks the host tables chunk by chunk
    const uint64_t n_pairs_next = 1ULL << (p->nv - (p->round + 1));
    if (!(n_pairs_next <= small_pairs_limit() && p->U <= (uint32_t)scd::kMaxSmallTables && p->K > 0)) return false;
    return ensure_mailbox(p);
}
// the challenge of the round enqueued with deferred = true: mailbox first, then the signal the stream is waiting on
static void provide_challenge(sc_prover *p, const sch::Fr &r) {
    p->randomness.push_back(r);
    FrHost *slot = p->h_mail + (p->sig_seq & 1u);
    std::memcpy(slot, &r, sizeof(FrHost));
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __atomic_store_n(p->sig, p->sig_seq, __ATOMIC_RELEASE);
    p->deferred_pending = false;
}
// the give-up marker of k_wait_challenge (non-zero once any wait of this handle has expired; cleared by sc_prover_reset)
static bool wait_gave_up(sc_prover *p) { return p->sig && __atomic_load_n(p->sig + 1, __ATOMIC_ACQUIRE) != 0; }
// error path: let a stream that is blocked on the wait drain (the round then runs on a stale challenge; its result is discarded)
static void abandon_deferred(sc_prover *p) {
    DeviceGate gate_(p->device);
    if (p->deferred_pending) {
        __atomic_store_n(p->sig, p->sig_seq, __ATOMIC_RELEASE);
        p->deferred_pending = false;
        (void)hipStreamSynchronize(p->stream);
        p->exhausted = true; // tables are no longer meaningful: the handle must be reset
    }
}

// rows (r * 2^(29 i + 58)) mod p as plain 29-bit limbs: the challenge as the tree kernels' bind takes it (fe_device.hpp, fe_mul_bind)
static void make_bind_const(const sch::Fr &r, scd::BindConst &rc) {
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
static int launch_round_streamed(sc_prover *p, const uint64_t *r_or_null, uint64_t *d_wide, bool publish_to_host) {
    if (p->exhausted) return fail(SC_ERR_NOT_ACTIVE, "Prover is not active");
    if (r_or_null && p->round == 0) return fail(SC_ERR_FIRST_ROUND_HAS_MSG, "first round should be prover first.");
    if (!r_or_null && p->round > 0) return fail(SC_ERR_MISSING_MSG, "verifier message is empty");
    sch::Fr r = sch::zero();
    if (r_or_null) {
        std::memcpy(&r, r_or_null, 32);
        if (sch::geq_p(r)) return fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
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
static int launch_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *d_wide, bool publish_to_host, bool deferred = false) {
    if (p->res.active) { // (sc_prove_round_partial after interactive rounds)
        int rc_q = resident_quiesce(p);
        if (rc_q) return rc_q;
    }
    DeviceGate gate(p->device);
    if (p->streamed && p->round < 2 && !p->exhausted) { // the inputs are still in host memory: the round is computed chunk by chunk
        if (deferred) return fail(SC_ERR_BAD_ARG, "streamed tables: rounds 1 and 2 are not pipelined");
        return launch_round_streamed(p, r_or_null, d_wide, publish_to_host);
    }
    // validation, same precedence as the reference's panics (prover.rs:78-98)
    if (p->exhausted) return fail(SC_ERR_NOT_ACTIVE, "Prover is not active");
    if (p->deferred_pending) return fail(SC_ERR_BAD_ARG, "a pipelined round is waiting for its challenge");
    if (r_or_null && p->round == 0) return fail(SC_ERR_FIRST_ROUND_HAS_MSG, "first round should be prover first.");
    if (!r_or_null && p->round > 0 && !deferred) return fail(SC_ERR_MISSING_MSG, "verifier message is empty");
    if (p->round + 1 > p->nv) return fail(SC_ERR_NOT_ACTIVE, "Prover is not active");
    sch::Fr r = sch::zero();
    if (r_or_null) {
        std::memcpy(&r, r_or_null, 32);
        if (sch::geq_p(r)) return fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
    }
    if (deferred) { // (every check comes before the first change to the handle)
        const uint64_t np = 1ULL << (p->nv - (p->round + 1));
        if (!(np <= small_pairs_limit() && p->U <= (uint32_t)scd::kMaxSmallTables && p->K > 0)) return fail(SC_ERR_BAD_ARG, "only late rounds are pipelined");
    }
    HIP_TRY(hipSetDevice(p->device));
    if (!deferred) { // (a pipelined round records no events: collecting would wait for the round before it)
        int rc_t = collect_timing(p);
        if (rc_t) return rc_t;
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
    const bool small_round = n_pairs <= small_pairs && p->U <= (uint32_t)scd::kMaxSmallTables && p->K > 0;
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
        // latency-bound round: one launch binds every table, one launch sums every (product, point) combination
        TablePtrs tp;
        std::memset(&tp, 0, sizeof(tp));
        if (bind) {
            for (uint32_t u = 0; u < p->U; ++u) {
                Table &t = p->tabs[u];
                tp.src[u] = t.cur;
                tp.src_top[u] = t.cur_top;
                tp.dst[u] = t.buf[t.next];
            }
            {
                SlowCallProbe pr("launch k_fix_multi");
                HIP_TRY(scd::launch_fix_multi(tp, (int)p->U, rdev, r_mail, 2 * n_pairs, p->stream));
            }
            for (uint32_t u = 0; u < p->U; ++u) {
                Table &t = p->tabs[u];
                t.cur = t.buf[t.next];
                t.cur_top = nullptr; // the latency-bound path keeps tables canonical in the reference layout
                t.next ^= 1;
            }
        }
        for (uint32_t u = 0; u < p->U; ++u) tp.src[u] = p->tabs[u].cur;
        SlowCallProbe pr_sum("launch k_sum_combos");
        if (p->has_meta) HIP_TRY(scd::launch_sum_combos_meta(tp, p->meta, p->n_combos, n_pairs, p->d_partials, grid, p->stream));
        else HIP_TRY(scd::launch_sum_combos(tp, p->d_combos, p->n_combos, p->d_slot_table, p->d_slot_exp, n_pairs, p->d_partials, grid, p->stream));
        scaled = 1; // products of up to kMaxFusedM multiplicands are summed in carry-free arithmetic (2^261 radix) there too
        bind = false;
    }
    if (bind && p->any_generic) { // generic products read bound tables: bind everything up front
        for (uint32_t u = 0; u < p->U; ++u) HIP_TRY(bind_table(u));
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
        if (pr.fused && p->kernel_variant == 3 && pr.M <= 4) {
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
                HIP_TRY(scd::launch_prod_round_fe((int)pr.M, a, r32, n_pairs, partials, grid, p->stream));
                scaled = 1;
            }
        } else {
            if (!ptrs_uploaded) {
                for (uint32_t u = 0; u < p->U; ++u) p->h_cur_tables[u] = p->tabs[u].cur;
                HIP_TRY(hipMemcpyAsync(p->d_cur_tables, p->h_cur_tables, p->U * sizeof(void *), hipMemcpyHostToDevice, p->stream));
                ptrs_uploaded = true;
            }
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
                                 publish_to_host ? p->h_out_dev : nullptr, publish_to_host ? p->h_flag_dev : nullptr, p->seq, scaled,
                                 fin_mb_enabled() ? p->d_fin_mb_counter : nullptr, p->stream, skip1 ? &ca : nullptr));
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

static int await_round(sc_prover *p, uint64_t *out_evals, uint32_t want);

constexpr int kResidentGone = -1; // internal: no resident kernel serves this round; take the ordinary path
static int resident_start(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals);
static int resident_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals);
extern "C" int sc_prove_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals) {
    if (!p || !out_evals) return fail(SC_ERR_BAD_ARG, "null argument");
    // late rounds of the interactive protocol: a kernel that stays on the GPU between calls (see resident_start)
    int rc = p->res.active ? resident_round(p, r_or_null, out_evals) : resident_start(p, r_or_null, out_evals);
    if (rc != kResidentGone) return rc;
    rc = launch_round(p, r_or_null, nullptr, true);
    if (rc) return rc;
    return await_round(p, out_evals, p->seq);
}

// want: the sequence number the awaited round's finalize publishes (p->seq right after that round was launched)
static int await_round(sc_prover *p, uint64_t *out_evals, const uint32_t want) {
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
            return fail(SC_ERR_HIP, "round did not publish its message within 2 s");
        }
        {
            DeviceGate gate_(p->device);
            HIP_TRY(hipStreamSynchronize(p->stream));
        }
        if (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) != want) return fail(SC_ERR_HIP, "round finished without publishing its message");
    }
    if (wait_gave_up(p)) { // a wait kernel's bound expired before its challenge arrived: that round ran on a stale one
        if (std::getenv("SC_HOST_TRACE"))
            std::fprintf(stderr, "[sc] give-up seen in await_round: marker %u, sig word %u, sig_seq %u, awaited seq %u, h_flag %u, round %u, deferred_pending %d, waited %.3f s\n",
                         __atomic_load_n(p->sig + 1, __ATOMIC_ACQUIRE), __atomic_load_n(p->sig, __ATOMIC_ACQUIRE), p->sig_seq, want,
                         __atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE), p->round, (int)p->deferred_pending,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        abandon_deferred(p);
        p->exhausted = true;
        return fail(SC_ERR_HIP, "the host took longer than the wait kernel's bound to deliver a challenge; the proof is void");
    }
    std::memcpy(out_evals, p->h_out, (size_t)p->D * 32);
    return SC_OK;
}

// ---- the persistent tail: every remaining latency-bound round in ONE kernel launch (kernels.hip: k_tail_rounds) -------------
// Usable when the round metadata fits kernel arguments (tail_shape_ok), the next round is a small one, and launches are
// asynchronous (the kernel waits for the host; SC_PIPELINE=0 switches it off together with the pipelined rounds).
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
static std::atomic<uint64_t> g_stat[8];
enum { kStatTailLaunches = 0, kStatTailSlotBusy = 1, kStatTailSlotReclaims = 2, kStatResidentStarts = 3, kStatResidentGone = 4, kStatProofRetries = 5 };
struct TailOwner {
    std::mutex mu;
    sc_prover *owner = nullptr;
    bool resident = false;           // held by resident_start (reclaimable once the kernel has left)
    const uint32_t *marker = nullptr; // the holder's sig + 1
};
static TailOwner g_tail_owner[64];
static bool tail_slot_acquire(sc_prover *p, bool resident) {
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
static void tail_slot_release(sc_prover *p) { // (a holder that lost the slot to a reclaim releases nothing)
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
static bool tail_shape_ok(const sc_prover *p) {
    return p->use_tail && p->K > 0 && p->U <= (uint32_t)scd::kMaxSmallTables && p->has_meta && p->K <= (uint32_t)scd::kMetaProds &&
           (size_t)p->K * p->D * (p->D + 2) * 32 <= 48 * 1024;
}
static bool tail_possible(sc_prover *p) {
    if (!tail_shape_ok(p) || p->exhausted || p->round >= p->nv || p->deferred_pending) return false;
    if (p->streamed && p->round < 2) return false;
    if ((1ULL << (p->nv - (p->round + 1))) > std::min<uint64_t>(small_pairs_limit(), scd::kTailMaxPairs)) return false;
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
static int tail_launch(sc_prover *p, uint32_t n_rounds, const sch::Fr *r_or_null, uint32_t max_spins, scd::TailArgs &A, int &grid) {
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
    HIP_TRY(scd::launch_zero_words(p->d_tail_sync, (uint32_t)((kTailSyncBytes + 64) / 4), p->stream));
    HIP_TRY(scd::launch_tail_rounds(A, p->meta, fm, grid, p->stream));
    return SC_OK;
}
// where the tables are after a tail kernel that did `nb` binds
static void tail_epilogue_tables(sc_prover *p, uint32_t nb) {
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
static void tail_post_challenge(sc_prover *p, uint32_t sv, const sch::Fr &vm) {
    uint64_t *slot = reinterpret_cast<uint64_t *>(p->h_mail) + 16 + 8 * (sv & 1u);
    for (int i = 0; i < 8; ++i) {
        const uint32_t limb = (uint32_t)(vm.l[i >> 1] >> (32 * (i & 1)));
        __atomic_store_n(slot + i, ((uint64_t)limb << 32) | sv, __ATOMIC_RELEASE);
    }
}

// n_rounds rounds (prove_round, feed, sample) starting at the handle's next round; r_or_null = the challenge that round binds
static int run_tail(sc_prover *p, sch::Blake2b512Rng &rng, uint32_t n_rounds, const sch::Fr *r_or_null, uint64_t *out_msgs, sch::Fr *out_challenges) {
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
    int rc_l = tail_launch(p, n_rounds, r_or_null, scd::wait_spins_default(), A, grid);
    if (rc_l) return rc_l;
    g_stat[kStatTailLaunches].fetch_add(1, std::memory_order_relaxed);
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
                rc = fail(SC_ERR_HIP, wait_gave_up(p) ? "the host took longer than the wait kernel's bound to deliver a challenge; the proof is void"
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
            tail_post_challenge(p, A.sig0 + j + 1, vm);
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
static void resident_release_slot(sc_prover *p) { tail_slot_release(p); }
// the kernel has exited (all rounds done, patience expired, or stop word): fold what it did into the handle
static int resident_finish(sc_prover *p) {
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
        return fail(SC_ERR_HIP, "the resident round kernel failed: %s", hipGetErrorString(e));
    }
    if (r.done == 0) { // it never published: the tables may be half bound
        p->exhausted = true;
        return fail(SC_ERR_HIP, "the resident round kernel left before its first message");
    }
    tail_epilogue_tables(p, r.done - 1 + (r.first_has_bind ? 1 : 0));
    p->timed = false;
    return SC_OK;
}
// ask it to leave (any entry point other than sc_prove_round), then fold
static int resident_quiesce(sc_prover *p) {
    if (!p || !p->res.active) return SC_OK;
    if (p->res.done < p->res.n_rounds) { // it is (or will be) polling for the challenge tagged sig0 + done: word 0 with the stop bit
        const uint32_t want = p->res.sig0 + p->res.done;
        uint64_t *slot = reinterpret_cast<uint64_t *>(p->h_mail) + 16 + 8 * (want & 1u);
        __atomic_store_n(slot, (uint64_t)(want ^ 0x80000000u), __ATOMIC_RELEASE);
    }
    return resident_finish(p);
}
// wait for the message of the kernel's round j.  SC_OK: in out_evals; kResidentGone: the kernel left before computing it
static int resident_wait(sc_prover *p, uint32_t j, uint64_t *out_evals) {
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
                return fail(SC_ERR_HIP, "the resident round kernel did not publish its message within 20 s");
            }
        }
    }
    std::memcpy(out_evals, p->h_out, (size_t)p->D * 32);
    p->res.done = j + 1;
    p->round += 1;
    if (p->res.done == p->res.n_rounds) return resident_finish(p); // the last round: the kernel ends by itself
    return SC_OK;
}
static bool resident_enabled(sc_prover *p) {
    static const bool env_on = !(std::getenv("SC_RESIDENT") && std::atoi(std::getenv("SC_RESIDENT")) == 0);
    return env_on && p->resident_spins > 0 && !p->timing && p->stream == p->own_stream;
}
// first late-round call: launch the kernel for every remaining round, return its first message
static int resident_start(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals) {
    // argument errors keep the reference's precedence: the ordinary path reports them
    if ((r_or_null && p->round == 0) || (!r_or_null && p->round > 0)) return kResidentGone;
    sch::Fr r = sch::zero();
    if (r_or_null) {
        std::memcpy(&r, r_or_null, 32);
        if (sch::geq_p(r)) return kResidentGone;
    }
    if (!resident_enabled(p) || !tail_possible(p)) return kResidentGone;
    if (!tail_slot_acquire(p, true)) return kResidentGone;
    const uint32_t n_rounds = p->nv - p->round;
    scd::TailArgs A;
    int grid = 1;
    {
        DeviceGate gate_(p->device);
        int rc = hipSetDevice(p->device) == hipSuccess ? tail_launch(p, n_rounds, r_or_null ? &r : nullptr, p->resident_spins, A, grid) : SC_ERR_HIP;
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
        return fail(SC_ERR_HIP, "the resident round kernel left before its first message");
    }
    return rc;
}
// a following call: post the challenge, take the next message
static int resident_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals) {
    if (!r_or_null) return fail(SC_ERR_MISSING_MSG, "verifier message is empty"); // (round > 0 here; the kernel keeps waiting)
    sch::Fr r;
    std::memcpy(&r, r_or_null, 32);
    if (sch::geq_p(r)) return fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
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
static int run_rounds(sc_prover *p, sch::Blake2b512Rng &rng, uint32_t n_rounds, uint64_t *out_msgs, sch::Fr *out_challenges_or_null,
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
        if (!enqueued && tail_possible(p)) { // from here on every round is latency-bound: one persistent kernel runs them all
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
        const bool next_is_tail = tail_shape_ok(p) && p->round < p->nv && (1ULL << (p->nv - (p->round + 1))) <= scd::kTailMaxPairs &&
                                  !(p->streamed && p->round < 2);
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
    if (!p || !d_wide_out) return fail(SC_ERR_BAD_ARG, "null argument");
    return launch_round(p, r_or_null, d_wide_out, false);
}

// Bind the challenge `r` into every table once more and write the results back to back (table u at d_out + u * n * 4 limbs, n =
// 2^(num_vars - round) entries each, canonical form whatever the tables' internal format).  After this the handle is exhausted.
static int prover_bind_out(sc_prover *p, const uint64_t *r, uint64_t *d_out) {
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    DeviceGate gate_(p->device);
    if (p->exhausted || p->round == 0 || p->round > p->nv) return fail(SC_ERR_NOT_ACTIVE, "bind needs a prover that has run at least one round");
    if (p->streamed && p->round < 2) return fail(SC_ERR_NOT_ACTIVE, "a streamed handle's tables are resident from round 2 on: nothing to bind yet");
    sch::Fr rr;
    std::memcpy(&rr, r, 32);
    if (sch::geq_p(rr)) return fail(SC_ERR_BAD_ARG, "challenge is not a canonical field element");
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
    if (!p || !r || !d_out) return fail(SC_ERR_BAD_ARG, "null argument");
    if (p->exhausted || p->round != p->nv) return fail(SC_ERR_NOT_ACTIVE, "bind_final needs a prover that has finished its last local round");
    return prover_bind_out(p, r, d_out);
}

extern "C" int sc_prover_push_randomness(sc_prover *p, const uint64_t *r) {
    if (!p || !r) return fail(SC_ERR_BAD_ARG, "null argument");
    sch::Fr rr;
    std::memcpy(&rr, r, 32);
    p->randomness.push_back(rr);
    return SC_OK;
}

extern "C" int sc_prover_state(sc_prover *p, uint64_t *randomness, uint32_t *n_randomness, uint64_t *tables_out, uint32_t *round) {
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
    if (tables_out) {
        int rc_q = resident_quiesce(p);
        if (rc_q) return rc_q;
    }
    DeviceGate gate_(p->device);
    if (randomness && !p->randomness.empty()) std::memcpy(randomness, p->randomness.data(), p->randomness.size() * 32);
    if (n_randomness) *n_randomness = (uint32_t)p->randomness.size();
    if (round) *round = p->round;
    if (tables_out) {
        if (p->exhausted) return fail(SC_ERR_NOT_ACTIVE, "tables were consumed by sc_prover_bind_final");
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
    if (!p || !ms) return fail(SC_ERR_BAD_ARG, "null argument");
    if (!p->timed) return fail(SC_ERR_BAD_ARG, "no timed round: enable sc_prover_set_timing before the round");
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(ms, p->ev0, p->ev1));
    return SC_OK;
}

extern "C" int sc_prover_set_timing(sc_prover *p, int on) {
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
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
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
    DeviceGate gate_(p->device);
    if (!p->timing) return fail(SC_ERR_BAD_ARG, "timing is not enabled on this handle");
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
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
    DeviceGate gate_(p->device);
    if (!p->timing) return fail(SC_ERR_BAD_ARG, "timing is not enabled on this handle");
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
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
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
    const uint64_t n = 1ULL << p->nv;
    if (p->streamed) {
        if (flags & SC_TABLES_ON_DEVICE) return fail(SC_ERR_BAD_ARG, "streamed tables are host tables");
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (uint32_t u = 0; u < p->U; ++u) {
            if (tables_or_null) {
                if (!tables_or_null[u]) return fail(SC_ERR_BAD_ARG, "table %u is null", u);
                p->host_tabs[u] = tables_or_null[u];
            }
            p->tabs[u].cur = nullptr;
            p->tabs[u].cur_top = nullptr;
            p->tabs[u].next = 0;
        }
    } else if (p->borrow) {
        for (uint32_t u = 0; u < p->U; ++u) {
            if (tables_or_null) {
                if (!tables_or_null[u]) return fail(SC_ERR_BAD_ARG, "table %u is null", u);
                p->origin[u] = reinterpret_cast<const uint4 *>(tables_or_null[u]);
            }
            p->tabs[u].cur = p->origin[u];
            p->tabs[u].cur_top = nullptr;
            p->tabs[u].next = 0;
        }
    } else {
        if (!tables_or_null && p->host_tabs.size() == p->U) tables_or_null = p->host_tabs.data(); // sc_prover_init_streamed below its threshold
        if (!tables_or_null) return fail(SC_ERR_BAD_ARG, "a copying handle needs the tables again to reset");
        const bool on_device = flags & SC_TABLES_ON_DEVICE;
        if (on_device) HIP_TRY(hipDeviceSynchronize()); // the producer of the new tables may still be running on another stream
        for (uint32_t u = 0; u < p->U; ++u) {
            if (!tables_or_null[u]) return fail(SC_ERR_BAD_ARG, "table %u is null", u);
            HIP_TRY(hipMemcpyAsync(p->tabs[u].buf[0], tables_or_null[u], n * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                   p->stream));
            p->tabs[u].cur = p->tabs[u].buf[0];
            p->tabs[u].cur_top = nullptr;
            p->tabs[u].next = 1;
        }
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    p->round = 0;
    p->sums_round = -1;
    p->exhausted = false;
    p->randomness.clear();
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
    if (!in || !out || (k && !point)) return fail(SC_ERR_BAD_ARG, "null argument");
    if (k > nv || nv > 40) return fail(SC_ERR_BAD_ARG, "invalid partial point dimension"); // ark-poly's assert
    std::vector<sch::Fr> pt(k);
    for (uint32_t i = 0; i < k; ++i) {
        std::memcpy(&pt[i], point + 4 * i, 32);
        if (sch::geq_p(pt[i])) return fail(SC_ERR_BAD_ARG, "point[%u] is not a canonical field element", i);
    }
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
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
    if (!d || !out_value || (d->num_vars && !point)) return fail(SC_ERR_BAD_ARG, "null argument");
    if (d->num_vars > 40) return fail(SC_ERR_BAD_ARG, "num_vars %u too large", d->num_vars);
    if (d->n_tables == 0 || !d->tables) return fail(SC_ERR_BAD_ARG, "no tables");
    if (d->n_products && (!d->coeffs || !d->prod_offsets || !d->prod_indices)) return fail(SC_ERR_BAD_ARG, "null product arrays");
    for (uint32_t k = 0; k < d->n_products; ++k) {
        if (d->prod_offsets[k + 1] <= d->prod_offsets[k]) return fail(SC_ERR_BAD_ARG, "product %u is empty", k);
        for (uint32_t q = d->prod_offsets[k]; q < d->prod_offsets[k + 1]; ++q)
            if (d->prod_indices[q] >= d->n_tables) return fail(SC_ERR_BAD_ARG, "product %u refers to table %u >= %u", k, d->prod_indices[q], d->n_tables);
    }
    for (uint32_t u = 0; u < d->n_tables; ++u)
        if (!d->tables[u]) return fail(SC_ERR_BAD_ARG, "table %u is null", u);
    const uint32_t nv = d->num_vars, U = d->n_tables;
    std::vector<sch::Fr> pt(nv);
    for (uint32_t i = 0; i < nv; ++i) {
        std::memcpy(&pt[i], point + 4 * i, 32);
        if (sch::geq_p(pt[i])) return fail(SC_ERR_BAD_ARG, "point[%u] is not a canonical field element", i);
    }
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
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
        if (sch::geq_p(pr)) return fail(SC_ERR_BAD_ARG, "coefficient %u is not a canonical field element", k);
        for (uint32_t q = d->prod_offsets[k]; q < d->prod_offsets[k + 1]; ++q) pr = sch::mul(pr, tv[d->prod_indices[q]]);
        acc = sch::add(acc, pr);
    }
    std::memcpy(out_value, &acc, 32);
    if (out_table_values_or_null) std::memcpy(out_table_values_or_null, tv.data(), (size_t)U * 32);
    return SC_OK;
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
    if (!p || !rng || !out_proof || !out_randomness) return fail(SC_ERR_BAD_ARG, "null argument");
    if (p->round != 0 || n_rounds > p->nv) return fail(SC_ERR_BAD_ARG, "handle must be at round 0 and hold at least n_rounds variables");
    std::vector<sch::Fr> ch(n_rounds);
    int rc = sc_internal_run_rounds(p, rng->rng, n_rounds, out_proof, ch.data());
    if (rc) return rc;
    if (n_rounds) std::memcpy(out_randomness, ch.data(), (size_t)n_rounds * 32);
    return SC_OK;
}

extern "C" int sc_ml_prove_handle(sc_prover *p, sc_rng *rng_or_null, uint64_t *out_proof) {
    if (!p || !out_proof) return fail(SC_ERR_BAD_ARG, "null argument");
    if (p->round != 0) return fail(SC_ERR_BAD_ARG, "handle is not at round 0");
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
static std::vector<uint8_t> pool_key_of(const sc_poly_desc *d, int device) {
    std::vector<uint8_t> k;
    auto put = [&](const void *p, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        k.insert(k.end(), b, b + n);
    };
    const uint32_t head[6] = {d->num_vars, d->max_multiplicands, d->n_products, d->n_tables, d->flags, (uint32_t)device};
    put(head, sizeof(head));
    if (d->n_products) {
        put(d->prod_offsets, (size_t)(d->n_products + 1) * 4);
        put(d->prod_indices, (size_t)d->prod_offsets[d->n_products] * 4);
        put(d->coeffs, (size_t)d->n_products * 32);
    }
    return k;
}
static sc_prover *handle_pool_take(const std::vector<uint8_t> &key) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (!g_pool.h || g_pool.h->pool_key != key) return nullptr;
    sc_prover *p = g_pool.h;
    g_pool.h = nullptr;
    return p;
}
static bool handle_pool_offer(sc_prover *p) {
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

extern "C" int sc_library_stats(uint64_t *out, uint32_t n) {
    if (!out) return fail(SC_ERR_BAD_ARG, "null argument");
    for (uint32_t i = 0; i < n; ++i) out[i] = i < 8 ? g_stat[i].load(std::memory_order_relaxed) : 0;
    return SC_OK;
}

extern "C" int sc_set_cache_limit(uint64_t bytes) {
    const uint64_t before = g_cache_limit.exchange(bytes);
    return bytes < before ? sc_release_caches() : SC_OK;
}

extern "C" int sc_prover_set_resident(sc_prover *p, uint32_t patience_polls) {
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    p->resident_spins = patience_polls;
    return SC_OK;
}

extern "C" int sc_prover_set_polling(sc_prover *p, int allow) {
    if (!p) return fail(SC_ERR_BAD_ARG, "null prover");
    if (p->deferred_pending) return fail(SC_ERR_BAD_ARG, "a pipelined round is waiting for its challenge");
    int rc_q = resident_quiesce(p);
    if (rc_q) return rc_q;
    p->pipeline_ok = allow != 0;
    p->polling_off_by_caller = allow == 0;
    return SC_OK;
}

extern "C" int sc_ml_prove(const sc_poly_desc *desc, sc_rng *rng_or_null, uint64_t *out_proof, sc_prover **out_state_or_null) {
    if (!desc || !out_proof) return fail(SC_ERR_BAD_ARG, "null argument");
    if (out_state_or_null) *out_state_or_null = nullptr;
    int rc = validate_desc(desc); // prover_init panics on a constant before anything is proved (prover.rs:50-52)
    if (rc) return rc;
    sc_poly_desc eff = *desc;
    // Without a state to hand back the prover does not outlive this call, and it never writes a caller's table: device tables are
    // read in place instead of being copied first (their producers are waited for, as a copy would).
    if (!out_state_or_null && (eff.flags & SC_TABLES_ON_DEVICE) && !(eff.flags & SC_TABLES_BORROW)) {
        if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
        DeviceGate gate_(g_device);
        HIP_TRY(hipSetDevice(g_device));
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

// ---------------------------------------------------------------------------------------------------
// verifier side (host; O(nv * deg) scalar work) -- reference src/ml_sumcheck/protocol/verifier.rs
// ---------------------------------------------------------------------------------------------------
static sch::Fr fr_from_u128(sch::u128 x) { // F::from(u128): lo + hi * 2^64
    const sch::Fr lo = sch::from_u64((uint64_t)x), hi = sch::from_u64((uint64_t)(x >> 64));
    static const sch::Fr two64 = sch::mul(sch::Fr{{0, 1, 0, 0}}, sch::kR2);
    return sch::add(lo, sch::mul(hi, two64));
}

// verifier.rs:139-251, literally: the value at x of the unique polynomial of degree < len through (i, y[i]), with the same
// three tiers for the ratio of denominators (machine i64 / i128 / field) and the same early returns.  Every tier computes
// the same field value; they are kept so that a reader can diff this against the reference line by line.
static sch::Fr interpolate(const sch::Fr *y, uint32_t len, const sch::Fr &x) {
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
static int require_canonical(const uint64_t *limbs, size_t n_elems, const char *what) {
    for (size_t i = 0; i < n_elems; ++i) {
        sch::Fr v;
        std::memcpy(&v, limbs + 4 * i, 32);
        if (sch::geq_p(v)) return fail(SC_ERR_BAD_ARG, "%s element %zu is not a canonical field element", what, i);
    }
    return SC_OK;
}

extern "C" int sc_interpolate_uni_poly(const uint64_t *p_i, uint32_t len, const uint64_t *eval_at, uint64_t *out) {
    if (!p_i || !eval_at || !out || len == 0) return fail(SC_ERR_BAD_ARG, "null argument");
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
    if (!claimed_sum || !proof || !out_point || !out_expected) return fail(SC_ERR_BAD_ARG, "null argument");
    // every message is read at [0] and [1] (verifier.rs:101-102; a one-element message makes the reference panic on the index): a
    // polynomial without multiplicands has no valid proof, and max_multiplicands + 1 must not wrap
    if (max_multiplicands == 0 || max_multiplicands == UINT32_MAX)
        return fail(SC_ERR_BAD_ARG, "max_multiplicands %u: a round message needs at least two evaluations", max_multiplicands);
    const uint32_t D = max_multiplicands + 1;
    // verifier.rs:60-62 panics on a message of the wrong length; here the caller states how many elements `proof` holds
    if (proof_elems != (uint64_t)num_vars * D) return fail(SC_ERR_BAD_ARG, "incorrect number of evaluations");
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
        if (!sch::eq(sch::add(ev[0], ev[1]), expected)) return fail(SC_ERR_REJECT, "Prover message is not consistent with the claim.");
        expected = interpolate(ev, D, rs[i]);
    }
    if (num_vars) std::memcpy(out_point, rs.data(), (size_t)num_vars * 32);
    std::memcpy(out_expected, expected.l, 32);
    return SC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Sharded rounds inside the library: one RCCL all-reduce per round on the handle's stream.
// RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch process that resolves to the copy torch already
// mapped), so the library itself has no link-time dependency on it.
// ---------------------------------------------------------------------------------------------------
namespace {
struct NcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
} // namespace
static NcclApi g_nccl;
static int nccl_load() {
    if (g_nccl.lib) return SC_OK;
    // A copy the process has already mapped (PyTorch-ROCm ships its own librccl) is the one to use: two RCCLs in one process is one too
    // many.  Nothing is promoted to the global namespace (RTLD_LOCAL): the entry points are taken with dlsym from this handle, and the
    // host process's own symbol resolution is left alone.
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(SC_ERR_HIP, "cannot load librccl: %s", dlerror());
    g_nccl.GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_nccl.CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_nccl.AllReduce = reinterpret_cast<decltype(&ncclAllReduce)>(dlsym(h, "ncclAllReduce"));
    g_nccl.AllGather = reinterpret_cast<decltype(&ncclAllGather)>(dlsym(h, "ncclAllGather"));
    g_nccl.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_nccl.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.AllReduce || !g_nccl.AllGather || !g_nccl.CommDestroy)
        return fail(SC_ERR_HIP, "librccl lacks the NCCL entry points");
    g_nccl.lib = h;
    return SC_OK;
}
// A communicator is either an RCCL one (collectives enqueued on the prover's stream, device buffers) or a HOST transport: two
// caller-supplied functions that exchange host buffers (MPI, gloo, shared memory between the threads of one process, ...).
// sc_comm_init_p2p: the ranks are threads of this process, one GPU each; they find each other in a process-wide registry under a group
// id of the caller's choosing.  The group holds every rank's inbox pointer and a small host barrier that also passes one pointer per
// rank around (the tail's gather buffers).
struct P2PGroup {
    std::mutex mu;
    std::condition_variable cv;
    int nranks = 0, joined = 0, left = 0;
    uint64_t *inbox[scd::kP2PMaxRanks] = {};
    int device[scd::kP2PMaxRanks] = {};
    // barrier + pointer exchange
    int arrived = 0;
    uint64_t phase = 0;
    void *ptrs[scd::kP2PMaxRanks] = {};
    bool broken = false;
    // every rank deposits `mine`, all leave with everybody's; false on timeout (the group is then unusable)
    bool exchange(int rank, void *mine, void **all_out) {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t my_phase = phase;
        ptrs[rank] = mine;
        if (++arrived == nranks) {
            arrived = 0;
            if (all_out) std::copy(ptrs, ptrs + nranks, all_out);
            last = std::vector<void *>(ptrs, ptrs + nranks);
            ++phase;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return phase != my_phase || broken; }) || broken) {
            broken = true;
            cv.notify_all();
            return false;
        }
        if (all_out) std::copy(last.begin(), last.end(), all_out);
        return true;
    }
    std::vector<void *> last;
};
static std::mutex g_p2p_mu;
static std::map<uint64_t, std::shared_ptr<P2PGroup>> g_p2p_groups;

struct sc_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    sc_allreduce_u64_fn h_allreduce = nullptr;
    sc_allgather_fn h_allgather = nullptr;
    void *ctx = nullptr;
    // peer-to-peer (sc_comm_init_p2p)
    std::shared_ptr<P2PGroup> p2p;
    uint64_t p2p_id = 0;
    uint32_t p2p_gen = 0;          // generations used so far
    bool p2p_shared_device = false; // two ranks on one GPU (functional tests): no kernel may wait long for another rank's kernel
    int device = 0;
};
#define NCCL_TRY(expr)                                                                                             \
    do {                                                                                                           \
        int r_ = (int)(expr);                                                                                         \
        if (r_ != 0) return fail(SC_ERR_HIP, "%s failed: %s", #expr, g_nccl.GetErrorString ? g_nccl.GetErrorString((ncclResult_t)r_) : "?"); \
    } while (0)

extern "C" int sc_comm_unique_id(uint8_t *out128) {
    if (!out128) return fail(SC_ERR_BAD_ARG, "null argument");
    int rc = nccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    NCCL_TRY(g_nccl.GetUniqueId(reinterpret_cast<ncclUniqueId *>(out128)));
    return SC_OK;
}
extern "C" int sc_comm_init(const uint8_t *id128, int rank, int nranks, sc_comm **out) {
    if (!id128 || !out || rank < 0 || rank >= nranks) return fail(SC_ERR_BAD_ARG, "bad argument");
    int rc = nccl_load();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(g_device));
    sc_comm *c = new (std::nothrow) sc_comm();
    if (!c) return fail(SC_ERR_OOM, "host allocation failed");
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    int r = (int)g_nccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) {
        delete c;
        return fail(SC_ERR_HIP, "ncclCommInitRank failed: %s", g_nccl.GetErrorString ? g_nccl.GetErrorString((ncclResult_t)r) : "?");
    }
    c->rank = rank;
    c->nranks = nranks;
    *out = c;
    return SC_OK;
}
extern "C" int sc_comm_init_host(int rank, int nranks, sc_allreduce_u64_fn allreduce, sc_allgather_fn allgather, void *ctx, sc_comm **out) {
    if (!out || rank < 0 || rank >= nranks || (nranks > 1 && (!allreduce || !allgather))) return fail(SC_ERR_BAD_ARG, "bad argument");
    sc_comm *c = new (std::nothrow) sc_comm();
    if (!c) return fail(SC_ERR_OOM, "host allocation failed");
    c->rank = rank;
    c->nranks = nranks;
    c->h_allreduce = allreduce;
    c->h_allgather = allgather;
    c->ctx = ctx;
    *out = c;
    return SC_OK;
}
// Peer-to-peer communicator for thread ranks (one host thread and one GPU per rank inside ONE process): no collective library.  Every
// rank allocates a fine-grained inbox on its own device, the ranks meet in the process-wide registry under `group_id` (any number
// not in use by another live group; the call blocks until all `nranks` threads have made it, 60 s at most), peer access is enabled
// between the devices, and from then on a round's all-reduce is one small kernel per rank (kernels.h: P2PArgs) and the tail's
// gather is peer copies.  Ranks may share a GPU (functional tests on a one-GPU box): the exchange kernel then gives up quickly when a
// peer's kernel has not run yet -- it may be queued behind this one -- and the host launches it again.
extern "C" int sc_comm_init_p2p(uint64_t group_id, int rank, int nranks, sc_comm **out) {
    if (!out || rank < 0 || rank >= nranks || nranks > scd::kP2PMaxRanks) return fail(SC_ERR_BAD_ARG, "bad argument");
    *out = nullptr;
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    HIP_TRY(hipSetDevice(g_device));
    std::shared_ptr<P2PGroup> g;
    {
        std::lock_guard<std::mutex> lk(g_p2p_mu);
        auto &slot = g_p2p_groups[group_id];
        if (!slot) {
            slot = std::make_shared<P2PGroup>();
            slot->nranks = nranks;
        }
        g = slot;
    }
    uint64_t *inbox = nullptr;
    {
        DeviceGate gate_(g_device);
        // fine-grained: peers' stores become visible to this device's polls without a kernel boundary
        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&inbox), scd::kP2PInboxWords * 8, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&inbox), scd::kP2PInboxWords * 8));
        }
        HIP_TRY(hipMemset(inbox, 0, scd::kP2PInboxWords * 8));
        HIP_TRY(hipDeviceSynchronize());
    }
    auto give_up = [&](int code, const char *msg) {
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->broken = true;
            g->cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(g_p2p_mu);
            auto it = g_p2p_groups.find(group_id);
            if (it != g_p2p_groups.end() && it->second == g) g_p2p_groups.erase(it);
        }
        (void)hipFree(inbox);
        return fail(code, "%s", msg);
    };
    {
        std::unique_lock<std::mutex> lk(g->mu);
        if (g->nranks != nranks || g->inbox[rank] || g->broken) {
            lk.unlock();
            (void)hipFree(inbox);
            return fail(SC_ERR_BAD_ARG, "p2p group %llu: rank %d joined twice, or the ranks disagree on the group's size", (unsigned long long)group_id, rank);
        }
        g->inbox[rank] = inbox;
        g->device[rank] = g_device;
        ++g->joined;
        g->cv.notify_all();
        if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->joined == g->nranks || g->broken; }) || g->broken) {
            lk.unlock();
            return give_up(SC_ERR_HIP, "p2p group: not every rank joined within 60 s");
        }
    }
    sc_comm *c = new (std::nothrow) sc_comm();
    if (!c) return give_up(SC_ERR_OOM, "host allocation failed");
    c->rank = rank;
    c->nranks = nranks;
    c->p2p = g;
    c->p2p_id = group_id;
    c->device = g_device;
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        if (g->device[q] == g_device) {
            c->p2p_shared_device = true;
            continue;
        }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g_device, g->device[q]) != hipSuccess || !can) {
            delete c;
            return give_up(SC_ERR_HIP, "p2p group: a peer device is not accessible from this one (no xGMI / PCIe peer access)");
        }
        const hipError_t e = hipDeviceEnablePeerAccess(g->device[q], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            delete c;
            return give_up(SC_ERR_HIP, "hipDeviceEnablePeerAccess failed");
        }
        (void)hipGetLastError();
    }
    // (a rank that shares its GPU makes the whole group cautious: every rank must agree on whether rounds are pipelined)
    void *all[scd::kP2PMaxRanks];
    if (!g->exchange(rank, c->p2p_shared_device ? (void *)1 : nullptr, all)) {
        delete c;
        return give_up(SC_ERR_HIP, "p2p group: a rank dropped out during set-up");
    }
    for (int q = 0; q < nranks; ++q) c->p2p_shared_device |= all[q] != nullptr;
    *out = c;
    return SC_OK;
}

// p2p: every rank's `bytes` from d_send into every rank's d_recv (rank order), by peer copies; returns when all pieces are in place
static int p2p_allgather(sc_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t s) {
    void *recv[scd::kP2PMaxRanks];
    HIP_TRY(hipStreamSynchronize(s)); // d_send is complete (and d_recv no longer read by this rank's earlier work)
    {
        GateYield yield_(c->device, true);
        if (!c->p2p->exchange(c->rank, d_recv, recv)) return fail(SC_ERR_HIP, "p2p group: a rank did not reach the gather");
    }
    for (int q = 0; q < c->nranks; ++q) {
        char *dst = static_cast<char *>(recv[q]) + (size_t)c->rank * bytes;
        if (c->p2p->device[q] == c->device) HIP_TRY(hipMemcpyAsync(dst, d_send, bytes, hipMemcpyDeviceToDevice, s));
        else HIP_TRY(hipMemcpyPeerAsync(dst, c->p2p->device[q], d_send, c->device, bytes, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(c->device, true);
        if (!c->p2p->exchange(c->rank, nullptr, nullptr)) return fail(SC_ERR_HIP, "p2p group: a rank did not finish the gather");
    }
    return SC_OK;
}
// p2p: any number of lanes summed in place over the group (the ranks read each other's buffers directly)
static int p2p_allreduce_table(sc_comm *c, uint64_t *d_lanes, size_t n_words, hipStream_t s) {
    void *all[scd::kP2PMaxRanks];
    uint64_t *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), n_words * 8));
    struct Free {
        void *p;
        ~Free() { (void)hipFree(p); }
    } free_tmp{tmp};
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(c->device, true);
        if (!c->p2p->exchange(c->rank, d_lanes, all)) return fail(SC_ERR_HIP, "p2p group: a rank did not reach the all-reduce");
    }
    scd::PeerLanes pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.n = c->nranks;
    for (int q = 0; q < c->nranks; ++q) pl.p[q] = static_cast<const uint64_t *>(all[q]);
    HIP_TRY(scd::launch_sum_peer_lanes(pl, n_words, tmp, s));
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(c->device, true); // nobody overwrites its lanes while a peer still reads them
        if (!c->p2p->exchange(c->rank, nullptr, nullptr)) return fail(SC_ERR_HIP, "p2p group: a rank did not finish the all-reduce");
    }
    HIP_TRY(hipMemcpyAsync(d_lanes, tmp, n_words * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

// Diagnostic: one all-reduce and one all-gather of known patterns over the communicator, checked on every rank.
extern "C" int sc_comm_selftest(sc_comm *c) {
    if (!c) return fail(SC_ERR_BAD_ARG, "null argument");
    const int G = c->nranks, n = 40;
    std::vector<uint64_t> lanes(n), gathered((size_t)n * G);
    for (int i = 0; i < n; ++i) lanes[i] = (uint64_t)(c->rank + 1) * (uint64_t)(i + 1) + ((uint64_t)(c->rank + 1) << 40);
    const std::vector<uint64_t> mine = lanes;
    if (c->comm) {
        HIP_TRY(hipSetDevice(g_device));
        uint64_t *d = nullptr, *dg = nullptr;
        HIP_TRY(hipMalloc(&d, n * 8));
        HIP_TRY(hipMalloc(&dg, (size_t)n * 8 * G));
        HIP_TRY(hipMemcpy(d, lanes.data(), n * 8, hipMemcpyHostToDevice));
        NCCL_TRY(g_nccl.AllGather(d, dg, (size_t)n, ncclUint64, c->comm, nullptr));
        NCCL_TRY(g_nccl.AllReduce(d, d, (size_t)n, ncclUint64, ncclSum, c->comm, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(lanes.data(), d, n * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(gathered.data(), dg, (size_t)n * 8 * G, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        (void)hipFree(dg);
    } else if (c->p2p) {
        HIP_TRY(hipSetDevice(c->device));
        DeviceGate gate_(c->device);
        uint64_t *d = nullptr, *dg = nullptr;
        HIP_TRY(hipMalloc(&d, n * 8));
        HIP_TRY(hipMalloc(&dg, (size_t)n * 8 * G));
        HIP_TRY(hipMemcpy(d, lanes.data(), n * 8, hipMemcpyHostToDevice));
        int rc = p2p_allgather(c, d, dg, (size_t)n * 8, nullptr);
        if (!rc) rc = p2p_allreduce_table(c, d, (size_t)n, nullptr);
        if (!rc) {
            HIP_TRY(hipMemcpy(lanes.data(), d, n * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(gathered.data(), dg, (size_t)n * 8 * G, hipMemcpyDeviceToHost));
        }
        {
            GateYield yield_(c->device, true); // (a peer may still be reading d: leave together)
            (void)c->p2p->exchange(c->rank, nullptr, nullptr);
        }
        (void)hipFree(d);
        (void)hipFree(dg);
        if (rc) return rc;
    } else if (G > 1) {
        GateYield yield_(g_device, true);
        if (c->h_allgather(c->ctx, mine.data(), gathered.data(), (size_t)n * 8) != 0) return fail(SC_ERR_HIP, "the host transport's all-gather failed");
        if (c->h_allreduce(c->ctx, lanes.data(), (size_t)n) != 0) return fail(SC_ERR_HIP, "the host transport's all-reduce failed");
    } else {
        gathered = mine;
    }
    const uint64_t tri = (uint64_t)G * (uint64_t)(G + 1) / 2;
    for (int i = 0; i < n; ++i) {
        if (lanes[i] != tri * (uint64_t)(i + 1) + (tri << 40)) return fail(SC_ERR_HIP, "all-reduce returned a wrong sum in word %d", i);
        for (int g = 0; g < G; ++g)
            if (gathered[(size_t)g * n + i] != (uint64_t)(g + 1) * (uint64_t)(i + 1) + ((uint64_t)(g + 1) << 40))
                return fail(SC_ERR_HIP, "all-gather returned a wrong word (rank %d, word %d)", g, i);
    }
    return SC_OK;
}
// gkr.hip: sum `n_words` uint64 lanes in device memory over the ranks of `comm`, in place; returns with the result visible on `s`
int sc_internal_allreduce_lanes(sc_comm *c, uint64_t *d_lanes, size_t n_words, hipStream_t s) {
    if (!c || c->nranks == 1) return SC_OK;
    if (c->comm) {
        NCCL_TRY(g_nccl.AllReduce(d_lanes, d_lanes, n_words, ncclUint64, ncclSum, c->comm, s));
        return SC_OK;
    }
    if (c->p2p) return p2p_allreduce_table(c, d_lanes, n_words, s);
    std::vector<uint64_t> h(n_words);
    HIP_TRY(hipMemcpyAsync(h.data(), d_lanes, n_words * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(g_device, true);
        if (c->h_allreduce(c->ctx, h.data(), n_words) != 0) return fail(SC_ERR_HIP, "the host transport's all-reduce failed");
    }
    HIP_TRY(hipMemcpyAsync(d_lanes, h.data(), n_words * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // `h` goes out of scope
    return SC_OK;
}
int sc_internal_comm_ranks(sc_comm *c) { return c ? c->nranks : 1; }

extern "C" int sc_comm_info(sc_comm *c, int *rank, int *nranks, int *kind) {
    if (!c) return fail(SC_ERR_BAD_ARG, "null argument");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (kind) *kind = c->comm ? SC_COMM_RCCL : c->p2p ? SC_COMM_P2P : SC_COMM_HOST;
    return SC_OK;
}
// Measurement, collective: `iters` back-to-back exchanges of n_words uint64 lanes in exactly the form a sharded round uses on this
// communicator -- RCCL: ncclAllReduce on a stream, the publishing kernel, the host's poll of the flag; peer-to-peer: the one exchange
// kernel and the poll; host transport: the publishing kernel, the poll, the caller's all-reduce function -- each waited for before
// the next is issued, as the rounds of a proof are.  The sums are checked.  *us_mean_out = wall time per exchange on this rank.
extern "C" int sc_comm_exchange_bench(sc_comm *c, uint32_t n_words, uint32_t iters, double *us_mean_out, double *us_min_out) {
    if (!c || !us_mean_out || n_words == 0 || n_words > (uint32_t)scd::kP2PWords || iters == 0) return fail(SC_ERR_BAD_ARG, "bad argument");
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    const int device = c->p2p ? c->device : g_device;
    HIP_TRY(hipSetDevice(device));
    struct Res {
        hipStream_t s = nullptr;
        uint64_t *d = nullptr, *h = nullptr, *h_dev = nullptr;
        uint32_t *flag = nullptr, *flag_dev = nullptr;
        ~Res() {
            if (s) (void)hipStreamSynchronize(s);
            if (d) (void)hipFree(d);
            if (h) (void)hipHostFree(h);
            if (flag) (void)hipHostFree(flag);
            if (s) (void)hipStreamDestroy(s);
        }
    } R;
    {
        DeviceGate gate_(device);
        HIP_TRY(hipStreamCreateWithFlags(&R.s, hipStreamNonBlocking));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&R.d), (size_t)n_words * 8));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&R.h), (size_t)n_words * 8, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&R.h_dev), R.h, 0));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&R.flag), 64, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&R.flag_dev), R.flag, 0));
        *R.flag = 0;
    }
    const bool p2p = c->p2p != nullptr && c->nranks > 1;
    std::vector<uint64_t> mine(n_words), lanes(n_words);
    double total_us = 0.0, min_us = 1e30;
    const uint64_t tri = (uint64_t)c->nranks * (uint64_t)(c->nranks + 1) / 2;
    for (uint32_t it = 0; it <= iters; ++it) { // (iteration 0 warms up and is not counted)
        for (uint32_t w = 0; w < n_words; ++w) mine[w] = (uint64_t)(c->rank + 1) * (uint64_t)(w + 1 + it);
        {
            DeviceGate gate_(device);
            HIP_TRY(hipMemcpyAsync(R.d, mine.data(), (size_t)n_words * 8, hipMemcpyHostToDevice, R.s));
            HIP_TRY(hipStreamSynchronize(R.s));
        }
        const uint32_t want = it + 1;
        scd::P2PArgs xa;
        const auto t0 = std::chrono::steady_clock::now();
        {
            DeviceGate gate_(device);
            if (c->comm) NCCL_TRY(g_nccl.AllReduce(R.d, R.d, (size_t)n_words, ncclUint64, ncclSum, c->comm, R.s));
            if (p2p) {
                std::memset(&xa, 0, sizeof(xa));
                for (int q = 0; q < c->nranks; ++q) xa.inbox[q] = c->p2p->inbox[q];
                xa.nranks = c->nranks;
                xa.rank = c->rank;
                xa.n_words = (int)n_words;
                xa.gen = ++c->p2p_gen;
                xa.max_spins = c->p2p_shared_device ? 2048u : scd::wait_spins_default();
                HIP_TRY(scd::launch_p2p_allreduce(xa, R.d, R.h_dev, R.flag_dev, want, R.s));
            } else {
                HIP_TRY(scd::launch_publish_words(R.d, R.h_dev, (int)n_words, R.flag_dev, want, R.s));
            }
        }
        uint64_t spins = 0;
        for (;;) {
            const uint32_t f = __atomic_load_n(R.flag, __ATOMIC_ACQUIRE);
            if (f == want) break;
            if (p2p && f == (want | scd::kP2PRetryBit)) { // (ranks sharing a GPU: a peer's kernel was queued behind this one)
                if (!c->p2p_shared_device) return fail(SC_ERR_HIP, "p2p all-reduce: a peer's lanes did not arrive");
                __atomic_store_n(R.flag, 0u, __ATOMIC_RELEASE);
                std::this_thread::yield();
                DeviceGate gate_(device);
                HIP_TRY(scd::launch_p2p_allreduce(xa, R.d, R.h_dev, R.flag_dev, want, R.s));
                continue;
            }
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > publish_timeout()) return fail(SC_ERR_HIP, "exchange %u did not publish", it);
        }
        std::copy(R.h, R.h + n_words, lanes.begin());
        if (!c->comm && !p2p && c->nranks > 1) {
            if (c->h_allreduce(c->ctx, lanes.data(), (size_t)n_words) != 0) return fail(SC_ERR_HIP, "the host transport's all-reduce failed");
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        for (uint32_t w = 0; w < n_words; ++w)
            if (lanes[w] != tri * (uint64_t)(w + 1 + it)) return fail(SC_ERR_HIP, "exchange %u returned a wrong sum in word %u", it, w);
        if (it > 0) {
            total_us += us;
            min_us = std::min(min_us, us);
        }
    }
    *us_mean_out = total_us / iters;
    if (us_min_out) *us_min_out = min_us;
    return SC_OK;
}

extern "C" void sc_comm_free(sc_comm *c) {
    if (!c) return;
    if (c->comm && g_nccl.CommDestroy) (void)g_nccl.CommDestroy(c->comm);
    if (c->p2p) { // the inboxes go when the LAST rank leaves: a peer's kernel may still be pushing into this one
        std::shared_ptr<P2PGroup> g = c->p2p;
        bool last = false;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            last = ++g->left == g->nranks;
        }
        if (last) {
            for (int q = 0; q < g->nranks; ++q)
                if (g->inbox[q]) {
                    (void)hipSetDevice(g->device[q]);
                    (void)hipDeviceSynchronize();
                    (void)hipFree(g->inbox[q]);
                }
            (void)hipSetDevice(g_device);
            std::lock_guard<std::mutex> lk(g_p2p_mu);
            auto it = g_p2p_groups.find(c->p2p_id);
            if (it != g_p2p_groups.end() && it->second == g) g_p2p_groups.erase(it);
        }
    }
    delete c;
}

// The first n_rounds rounds of MLSumcheck::prove_as_subprotocol (reference src/ml_sumcheck/mod.rs:54-64) on this rank's shard:
// per round the shard's kernels, one all-reduce (sum, uint64) of the (deg+1) x 8 zero-extended limbs -- ncclAllReduce on the same
// stream, or the host transport's function on the published lanes -- then, on every rank identically, fold, feed, sample.  The
// handle is left after round n_rounds (its tables have two entries when n_rounds == its num_vars).
static int sharded_rounds(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, uint32_t n_rounds, uint64_t *out_proof, uint64_t *out_randomness) {
    HIP_TRY(hipSetDevice(p->device));
    const int n_words = (int)p->D * 8;
    // (before anything changes the handle: the peer-to-peer inbox holds kP2PWords lanes per source, i.e. messages of at most 8 evaluations)
    if (comm->p2p && comm->nranks > 1 && n_words > scd::kP2PWords)
        return fail(SC_ERR_BAD_ARG, "a peer-to-peer communicator carries round messages of at most %d evaluations (max_multiplicands <= %d); this polynomial has %u",
                    scd::kP2PWords / 8, scd::kP2PWords / 8 - 1, p->D);
    if (!p->d_wide) {
        HIP_TRY(hipMalloc(&p->d_wide, (size_t)n_words * 8));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_wide), (size_t)n_words * 8, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_wide_dev), p->h_wide, 0));
    }
    const bool p2p = comm->p2p != nullptr && comm->nranks > 1;
    const bool on_stream = comm->comm != nullptr || p2p; // RCCL / p2p: the reduction is a stream operation between the round and its publication
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
    auto enqueue = [&](const uint64_t *r, bool deferred, uint32_t *want_out) -> int {
        DeviceGate gate_(p->device);
        int rc = launch_round(p, r, p->d_wide, false, deferred);
        if (rc) return rc;
        if (comm->comm) NCCL_TRY(g_nccl.AllReduce(p->d_wide, p->d_wide, (size_t)n_words, ncclUint64, ncclSum, comm->comm, p->stream));
        p->seq += 1;
        *want_out = p->seq;
        if (p2p) { // the all-reduce and the publication are one kernel
            fill_xargs(xargs[*want_out & 1u]);
            HIP_TRY(scd::launch_p2p_allreduce(xargs[*want_out & 1u], p->d_wide, p->h_wide_dev, p->h_flag_dev, *want_out, p->stream));
        } else {
            HIP_TRY(scd::launch_publish_words(p->d_wide, p->h_wide_dev, n_words, p->h_flag_dev, *want_out, p->stream));
        }
        return SC_OK;
    };
    for (uint32_t i = 0; i < n_rounds; ++i) {
        int rc;
        if (!enqueued && (rc = enqueue(have ? vm.l : nullptr, false, &want))) return rc;
        uint32_t want_next = 0;
        bool next_enqueued = false;
        if (i + 1 < n_rounds && may_defer && can_defer_next(p)) {
            gate_lock(p->device); // held until the challenge is handed over: see DeviceGate
            if ((rc = enqueue(nullptr, true, &want_next))) {
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
        while (!(seen = (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) == want))) {
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
            return fail(SC_ERR_HIP, "p2p all-reduce: a peer's lanes did not arrive");
        }
        if (!seen) {
            if (p->deferred_pending) {
                abandon_deferred(p);
                return fail(SC_ERR_HIP, "sharded round did not publish within 20 s");
            }
            HIP_TRY(hipStreamSynchronize(p->stream));
            if (__atomic_load_n(p->h_flag, __ATOMIC_ACQUIRE) != want) return fail(SC_ERR_HIP, "sharded round finished without publishing");
        }
        if (wait_gave_up(p)) {
            abandon_deferred(p);
            p->exhausted = true;
            return fail(SC_ERR_HIP, "the host took longer than the wait kernel's bound to deliver a challenge; the proof is void");
        }
        std::vector<uint64_t> lanes(p->h_wide, p->h_wide + n_words); // (the device reuses the page for the next round)
        if (!on_stream && comm->nranks > 1) {
            GateYield yield_(p->device, true);
            if (comm->h_allreduce(comm->ctx, lanes.data(), (size_t)n_words) != 0) {
                abandon_deferred(p);
                return fail(SC_ERR_HIP, "the host transport's all-reduce failed");
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
        // (`window` releases the gate here)
    }
    return SC_OK;
}

extern "C" int sc_ml_prove_sharded_rounds(sc_prover *p, sc_comm *comm, sc_rng *rng, uint32_t nv_total, uint32_t n_rounds, uint64_t *out_proof,
                                          uint64_t *out_randomness) {
    if (!p || !comm || !rng || !out_proof || !out_randomness) return fail(SC_ERR_BAD_ARG, "null argument");
    if (p->round != 0 || n_rounds > p->nv) return fail(SC_ERR_BAD_ARG, "handle must be at round 0 and hold at least n_rounds variables");
    rng->rng.feed_poly_info(p->max_mult, nv_total); // mod.rs:54: the GLOBAL instance's info
    return sharded_rounds(p, comm, rng->rng, n_rounds, out_proof, out_randomness);
}

// The tail of a sharded proof.  Sharded rounds pay a collective each; once the GLOBAL instance is down to a latency-bound size
// (at most 2^14 pairs) it is cheaper to stop sharding: every rank binds the last challenge into what is left of its shard (2^m
// entries per table), the remainders are all-gathered (U * 2^m * 32 bytes per rank), and every rank finishes the last m + log2 G
// rounds on the same complete tables -- no exchange any more (the transcripts stay in step: they absorb identical messages), at the
// single-GPU cost per round, in the persistent tail kernel.  m = 0 (one element per table and rank) is the smallest case.
static uint32_t sharded_tail_m(uint32_t nv_local, uint32_t k) { // log2 of the entries per table a rank still holds at the gather
    if (k == 0) return 0;
    const uint32_t want = k >= 15 ? 0u : 15u - k; // 2^(m + k - 1) pairs <= 2^14 in the first replicated round
    return std::min(want, nv_local - 1);          // at least one sharded round
}
static int sharded_tail(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, const uint64_t *last_challenge, uint32_t k, uint32_t m, uint64_t *out_proof,
                        uint64_t *out_randomness) {
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
                if (comm->h_allgather(comm->ctx, send.data(), recv.data(), send_bytes) != 0) return fail(SC_ERR_HIP, "the host transport's all-gather failed");
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
static int sharded_proof_body(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, uint32_t nv_total, uint64_t *out_proof, uint64_t *out_randomness) {
    const uint32_t G = (uint32_t)comm->nranks;
    if (G == 0 || (G & (G - 1)) != 0) return fail(SC_ERR_BAD_ARG, "the number of ranks must be a power of two");
    uint32_t k = 0;
    while ((1u << k) < G) ++k;
    if (p->round != 0 || p->nv + k != nv_total) return fail(SC_ERR_BAD_ARG, "handle must be at round 0 and hold a 1/%u shard of %u variables", G, nv_total);
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
    if (!p || !comm || !out_proof || !out_randomness) return fail(SC_ERR_BAD_ARG, "null argument");
    sc_rng local;
    sch::Blake2b512Rng &rng = rng_or_null ? rng_or_null->rng : local.rng;
    rng.feed_poly_info(p->max_mult, nv_total); // mod.rs:54: the GLOBAL instance's info
    return sharded_proof_body(p, comm, rng, nv_total, out_proof, out_randomness);
}
// gkr.hip: the rounds of one GKR sumcheck phase over a sharded pair of tables -- GKRRoundSumcheck::prove feeds no PolynomialInfo
// (gkr_round_sumcheck/mod.rs:108-133) -- and the communicator's shape
int sc_internal_sharded_phase(sc_prover *p, sc_comm *comm, sch::Blake2b512Rng &rng, uint32_t nv_total, uint64_t *out_proof, uint64_t *out_randomness) {
    if (!p || !comm || !out_proof || !out_randomness) return fail(SC_ERR_BAD_ARG, "null argument");
    return sharded_proof_body(p, comm, rng, nv_total, out_proof, out_randomness);
}
int sc_internal_comm_rank(sc_comm *c) { return c ? c->rank : 0; }

// ---------------------------------------------------------------------------------------------------
// integer all-reduce lanes -> field
// ---------------------------------------------------------------------------------------------------
extern "C" int sc_wide_reduce(const uint64_t *wide, uint32_t n_elems, uint64_t *out) {
    if (!wide || !out) return fail(SC_ERR_BAD_ARG, "null argument");
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
        if (v[5] != 0) return fail(SC_ERR_BAD_ARG, "wide lanes overflow");
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
    if (!d_out) return fail(SC_ERR_BAD_ARG, "null argument");
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible");
    HIP_TRY(hipSetDevice(g_device));
    HIP_TRY(scd::launch_synth(seed, stream, first, n, reinterpret_cast<uint4 *>(d_out), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return SC_OK;
}

extern "C" int sc_claim_weights(uint32_t M, const uint64_t *r, uint64_t *out) {
    if (!r || !out) return fail(SC_ERR_BAD_ARG, "null argument");
    if (M < 1 || M > 4) return fail(SC_ERR_BAD_ARG, "claim weights exist for 1..4 multiplicands, not %u", M);
    sch::Fr rr, lam[5];
    std::memcpy(&rr, r, 32);
    if (sch::geq_p(rr)) return fail(SC_ERR_BAD_ARG, "the point is not a canonical field element");
    claim_weights(M, rr, lam);
    std::memcpy(out, lam, (size_t)(M + 1) * 32);
    return SC_OK;
}

extern "C" int sc_fr_elementwise(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n) {
    if (!a || !b || !out) return fail(SC_ERR_BAD_ARG, "null argument");
    if (op < 0 || op > 5) return fail(SC_ERR_BAD_ARG, "unknown op %d", op);
    if (n == 0) return SC_OK;
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
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
    if (!ms_out) return fail(SC_ERR_BAD_ARG, "null argument");
    if (sc_device_count() <= 0) return fail(SC_ERR_HIP, "no HIP device visible");
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
