// prover_internal.hpp -- what the three translation units of the C ABI share: the prover handle, the communicator, the device gate,
// error plumbing.  abi.hip: handles (build / free / state / reset / pool), stand-alone operations (fix_variables, evaluate, verify);
// protocol.hip: the round launch plan and the protocol loops (prove_round, MLSumcheck::prove, the sharded proof); comm.hip: the three
// communicators (RCCL, host transport, peer-to-peer).  Nothing here is exported: the library is built with -fvisibility=hidden.
#pragma once
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

std::chrono::milliseconds publish_timeout(); // how long a host loop waits for a round's message before it declares the proof dead
// ---- the device gate (abi.hip) ----
extern thread_local uint16_t g_gate_depth[64];
void gate_lock(int device);
void gate_unlock(int device);
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
int sc_internal_fail(int code, const char *fmt, ...); // sets sc_last_error(), returns code
int &sc_internal_device_ref();                         // the calling thread's device (sc_set_device)
// (a failed HIP call also leaves its code as the thread's sticky "last error": it is taken out here, or the next kernel launch of this
// thread -- whose wrapper returns hipGetLastError() -- would report it again: one refused hipMalloc must not fail the proofs after it)
#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess) {                                                                                         \
            (void)hipGetLastError();                                                                                    \
            return sc_internal_fail(e_ == hipErrorOutOfMemory ? SC_ERR_OOM : SC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                                     \
        }                                                                                                               \
    } while (0)

static inline FrHost to_dev(const sch::Fr &a) {
    FrHost h;
    std::memcpy(&h, &a, sizeof(h));
    return h;
}

struct sc_rng {
    sch::Blake2b512Rng rng;
};

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
    FrHost *d_W = nullptr; // node -> message matrices of every product (see FinProd::w_off); w_elems of them, and the same again behind
    uint32_t w_elems = 0;  // (the descriptor's matrices: what a reset restores after sc_internal_scale_by_bound_table -- GKR phase two)
    bool w_scaled = false;
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
    uint32_t wide_gen = 0;           // ... with the tag of this generation of the communicator
    bool wide_tagged = false;        // (set by sharded_rounds for the duration of an RCCL proof with direct publication: finalize tags the lanes)
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
    uint64_t *d_tail_xw = nullptr;   // k_tail_slices: tagged hand-over words (kTsXwWords), and the next launch's first tag
    uint32_t ts_tag = 1;
    uint64_t *d_vram_mail = nullptr; // ... and its mailbox in (host-visible, fine-grained) device memory: two slots of eight tagged words
    int tail_max_blocks = 0;        // blocks of it the device holds at once (its grid never exceeds that)
    uint64_t arena_bytes = 0;       // size of the bound-table arena (what a pooled handle keeps allocated)
    std::vector<uint8_t> pool_key;  // non-empty: created by sc_ml_prove; sc_prover_free offers it back to the pool (handle_pool_*)
    uint32_t n_retries = 0;         // proofs repeated after an expired device-side wait (sc_ml_prove_handle)
    bool pipeline_ok = true;        // cleared when the wait-value path is unavailable (or sc_set_policy("pipeline", 0), SC_NO_DEVICE_POLLING, sc_prover_set_polling(p, 0))
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
    bool use_tail = true;           // sc_ml_prove* / GKR: the latency-bound rounds run in the persistent tail kernel (sc_set_policy("tail", 0): pipelined launches)
    bool merge_rounds = false; // big rounds run as ONE launch over all products (k_round_tree): <= kMaxRoundProds products of <= 4 multiplicands
    bool wide_tree = true; // products of 5..12 multiplicands as trees (policy "wide_tree" when the handle was built)
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
    hipStream_t copy_stream2 = nullptr;                      // staged initialisation: the odd tables' copies (a second transfer in flight hides the first one's start-up)
    hipEvent_t ev_copied2[2] = {nullptr, nullptr};
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    FrHost *d_chunk_msg = nullptr;        // a chunk's message, and the running sum over the chunks (2 x D elements)
    // staged initialisation (host tables copied in chunks, round 1 computed under the copy: staged_copy_and_round1): round 1's message is
    // already published under sequence number seq + 1, its node sums are in d_sums[1] if r1_keeps; the first launch_round only takes note
    bool r1_cached = false, r1_keeps = false;
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

// ---- abi.hip ----
int validate_desc(const sc_poly_desc *d);
void prover_destroy(sc_prover *p);
bool handle_pool_offer(sc_prover *p);
std::vector<uint8_t> pool_key_of(const sc_poly_desc *d, int device);
sc_prover *handle_pool_take(const std::vector<uint8_t> &key);
void claim_weights(uint32_t M, const sch::Fr &r, sch::Fr *lam);
int collect_timing(sc_prover *p);
int prover_bind_out(sc_prover *p, const uint64_t *r, uint64_t *d_out);
uint64_t sc_internal_cache_limit();
// ---- protocol.hip ----
constexpr int kResidentGone = -1; // internal: no resident kernel serves this round; take the ordinary path
extern std::atomic<uint64_t> g_stat[8]; // process-wide counters a host can read (sc_library_stats)
enum { kStatTailLaunches = 0, kStatTailSlotBusy = 1, kStatTailSlotReclaims = 2, kStatResidentStarts = 3, kStatResidentGone = 4, kStatProofRetries = 5, kStatTailSlices = 6 };
bool wide_tree_enabled(); // products of five to eight multiplicands through kernels_wide.hip (policy "wide_tree" = 0: node by node)
int resident_quiesce(sc_prover *p); // the interactive protocol's resident kernel leaves before anything else touches the handle
int launch_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *d_wide, bool publish_to_host, bool deferred = false);
bool staged_init_applies(const sc_prover *p);                                   // host tables, copy mode: the shape and size the staged form takes
int staged_copy_and_round1(sc_prover *p, const uint64_t *const *host_tables); // H2D in chunks + round 1 under the copy (prover.rs:55-59 and the first prove_round)
int await_round(sc_prover *p, uint64_t *out_evals, uint32_t want);
bool wait_gave_up(sc_prover *p);   // the give-up marker of k_wait_challenge
void abandon_deferred(sc_prover *p); // error path: let a stream that is blocked on the wait drain
int sc_internal_run_rounds(sc_prover *p, sch::Blake2b512Rng &rng, uint32_t n_rounds, uint64_t *out_msgs, sch::Fr *out_challenges);
// ---- comm.hip ----
struct NcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
extern NcclApi g_nccl;
int nccl_load();
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
    uint32_t direct_gen = 0;       // direct publication: all-reduces issued so far (every rank counts the same)
    bool direct_publish = false;   // RCCL: an all-reduce may deliver into host-mapped memory and the host sees tagged words there (probed at init)
    bool p2p_shared_device = false; // two ranks on one GPU (functional tests): no kernel may wait long for another rank's kernel
    int device = 0;
};
#define NCCL_TRY(expr)                                                                                             \
    do {                                                                                                           \
        int r_ = (int)(expr);                                                                                         \
        if (r_ != 0) return sc_internal_fail(SC_ERR_HIP, "%s failed: %s", #expr, g_nccl.GetErrorString ? g_nccl.GetErrorString((ncclResult_t)r_) : "?"); \
    } while (0)
int p2p_allgather(sc_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t s);
int p2p_allreduce_table(sc_comm *c, uint64_t *d_lanes, size_t n_words, hipStream_t s);
