/* sumcheck_hip.h -- C ABI of libsumcheck_hip.so: the MI355X (gfx950) prover hot path of the
 * linear-time multilinear sumcheck, a drop-in for arkworks-rs/sumcheck's
 *   IPForMLSumcheck::{prover_init, prove_round}      (reference src/ml_sumcheck/protocol/prover.rs:49,74)
 *   DenseMultilinearExtension::fix_variables          (ark-poly; called at prover.rs:88)
 *   MLSumcheck::{prove, prove_as_subprotocol}         (reference src/ml_sumcheck/mod.rs:42,50)
 *   gkr_round_sumcheck::{initialize_phase_one, initialize_phase_two, GKRRoundSumcheck::prove}
 *                                                      (reference src/gkr_round_sumcheck/mod.rs:22,57,93)
 *
 * The reference has no FFI of its own (it is pure Rust with #![forbid(unsafe_code)], src/lib.rs:1);
 * these entry points are what a Rust `extern "C"` shim crate binds (INTEGRATION.md shows the stub).
 *
 * ELEMENT LAYOUT.  A field element is `uint64_t[4]`: the little-endian limbs of the MONTGOMERY form
 * (R = 2^256) of a BLS12-381 scalar, always < p -- bit-for-bit the in-memory layout of
 * ark_ff::Fp<MontBackend<FrConfig,4>,4>, so `evaluations.as_ptr() as *const u64` can be passed
 * straight in and outputs can be wrapped back into `Fp` without conversion.  A table of 2^nv
 * evaluations is 2^nv * 32 contiguous bytes, index bit k <-> variable k (ark-poly
 * DenseMultilinearExtension).
 *
 * OWNERSHIP.  sc_prover_init copies its inputs (as prover_init deep-copies, prover.rs:55-59) unless
 * SC_TABLES_BORROW is set; it never writes caller memory.  Device tables that are COPIED (init or reset) are read after a
 * device-wide synchronise, so work still in flight on the caller's streams that produces them is waited for.  BORROWED device
 * tables are read by kernels on the handle's stream: the caller must have synchronised their producer before the first round.  A handle owns its device memory until
 * sc_prover_free.  Outputs go to caller-owned host buffers unless the parameter says "device".
 *
 * THREADING.  One handle = one host thread at a time.  Distinct handles are independent and may be used from different threads
 * concurrently, on one GPU or several; the library serialises its own HIP calls per device (a pipelined round must never find
 * the launch that follows its wait kernel blocked behind another thread's call), so threads sharing a GPU take turns at the
 * granularity of one round.  Calls are synchronous (return after the result is on the host) unless named *_async / *_partial.
 *
 * ERRORS.  Nothing aborts across the ABI.  The four misuse panics of the reference map to status codes
 * (the shim turns them back into the same panic! messages); sc_last_error() gives detail.
 * There is NO CPU fallback: without a usable HIP device every compute entry point returns SC_ERR_HIP.  A call that fails on a HIP error
 * (SC_ERR_OOM: an allocation was refused; SC_ERR_HIP) takes that error out of the runtime's per-thread "last error" slot, so it does not
 * resurface in the next call; the handle it failed on is freed (init) or must be reset (a proof in progress).
 */
#ifndef SUMCHECK_HIP_H
#define SUMCHECK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 5 /* 5: sc_set_policy, sc_get_policy, sc_plan_count, sc_plan_name, sc_plan_stats (additions only); 4: sc_comm_info, sc_comm_exchange_bench, sc_set_publish_timeout_ms, sc_prover_get_round_timing, sc_library_stats (additions only); 3: sc_prover_set_polling, sc_prover_set_resident, SC_NO_DEVICE_POLLING, sc_set_cache_limit, sc_comm_init_p2p, sc_gkr_prove_sharded (additions only) */
#define SC_API __attribute__((visibility("default")))

enum sc_status {
    SC_OK = 0,
    SC_ERR_CONSTANT_POLY = 1,       /* "Attempt to prove a constant."        prover.rs:50-52 */
    SC_ERR_FIRST_ROUND_HAS_MSG = 2, /* "first round should be prover first." prover.rs:79-81 */
    SC_ERR_MISSING_MSG = 3,         /* "verifier message is empty"           prover.rs:90-92 */
    SC_ERR_NOT_ACTIVE = 4,          /* "Prover is not active"                prover.rs:96-98 */
    SC_ERR_BAD_ARG = 5,             /* assert!/assert_eq! failures of data_structures.rs:78-84 and gkr mod.rs:28-29,61 */
    SC_ERR_HIP = 6,                 /* HIP runtime failure or no device */
    SC_ERR_OOM = 7,
    SC_ERR_REJECT = 8               /* verifier: "Prover message is not consistent with the claim." verifier.rs:107-113 */
};

enum sc_flags {
    SC_TABLES_ON_DEVICE = 1u << 0, /* `tables[]` are device pointers (HBM-resident), not host */
    SC_TABLES_BORROW = 1u << 1,    /* with ON_DEVICE: do not copy; tables are only read and must outlive the handle */
    SC_TABLES_STREAM = 1u << 2,    /* HOST tables too large to copy: see sc_prover_init_streamed (set by it; sc_prover_init accepts it too) */
    SC_NO_DEVICE_POLLING = 1u << 3 /* this handle never parks a kernel on the GPU that waits for the host (sc_prover_set_polling(p, 0) from the start) */
};

/* Flattened ListOfProductsOfPolynomials (reference src/ml_sumcheck/data_structures.rs:25-35):
 * products[k] = (coeffs[k], prod_indices[prod_offsets[k] .. prod_offsets[k+1]]) indexing `tables`. */
typedef struct sc_poly_desc {
    uint32_t num_vars;            /* num_variables */
    uint32_t max_multiplicands;   /* must equal max_k m_k (data_structures.rs:79) */
    uint32_t n_products;          /* K */
    const uint64_t *coeffs;       /* K x 4 limbs (host) */
    const uint32_t *prod_offsets; /* K+1 (host) */
    const uint32_t *prod_indices; /* sum m_k entries (host), each < n_tables */
    uint32_t n_tables;            /* U = flattened_ml_extensions.len() */
    const uint64_t *const *tables;/* U pointers, each 2^num_vars x 4 limbs */
    uint32_t flags;               /* sc_flags */
} sc_poly_desc;

typedef struct sc_prover sc_prover; /* ProverState (prover.rs:19-33) with tables resident in HBM */
typedef struct sc_rng sc_rng;       /* Blake2b512Rng (reference src/rng.rs:22-25) */

/* ---- library ------------------------------------------------------------------------------- */
SC_API int sc_abi_version(void);
SC_API const char *sc_last_error(void);     /* thread-local, valid until the next call on this thread */
SC_API int sc_device_count(void);           /* number of visible HIP devices (0 => every compute call fails) */
SC_API int sc_set_device(int ordinal);      /* device used by handles created afterwards on this thread */

/* ---- IPForMLSumcheck::prover_init / prove_round (prover.rs:49-153) ------------------------- */
SC_API int sc_prover_init(const sc_poly_desc *desc, sc_prover **out);
/* Out-of-core tables: the tables stay in HOST memory (pinned memory streams fastest; they are only read and must
 * stay valid until the second round has returned) and are never resident as a whole.  Rounds 1 and 2 pull them through a two-slot
 * staging ring, 2^chunk_log2 entries of every table at a time (0 = 2^22; clamped to [2^10, 2^num_vars]), the copy of one chunk
 * overlapping the kernels of the previous one; from round 2 on the bound tables -- half the input -- are resident and everything
 * proceeds as usual.  HBM footprint: 0.84 x the tables (bound-table buffers) + 2 x U x 2^chunk_log2 x 32 bytes, instead of 2.7 x.
 * The input crosses PCIe twice (rounds 1 and 2 both need it and the challenge between them comes from the verifier).
 * Any product shape: up to 12 products of at most 4 multiplicands walk a chunk in one launch (the merged big-round kernel); other
 * shapes bind the chunk table by table and sum product by product.  Below 2^11 entries per table the handle silently copies instead.
 * Every other call (sc_prove_round, sc_prove_round_partial / sc_ml_prove_sharded -- each rank of a multi-GPU proof may stream its own
 * shard --, sc_ml_prove_handle, sc_prover_state, sc_prover_reset with host tables or NULL) works on the handle unchanged. */
SC_API int sc_prover_init_streamed(const sc_poly_desc *desc, uint32_t chunk_log2, sc_prover **out);
/* r_or_null: NULL exactly on the first call, else the previous round's challenge (4 limbs).
 * out_evals: (max_multiplicands+1) x 4 limbs = [P(0), P(1), ..., P(deg)] (ProverMsg, prover.rs:13-17). */
SC_API int sc_prove_round(sc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals);
/* (Late rounds -- at most 2048 pairs -- of this round-by-round protocol do not pay a launch sequence each: the first of them launches
 * one kernel for all remaining rounds, which stays on the GPU polling a host-mapped mailbox; every following sc_prove_round posts
 * its challenge and takes the next message.  The kernel's patience is short, `patience_polls` polls of ~2 us (default 256; 0 = never
 * use a resident kernel): a verifier that takes longer finds it gone -- it leaves after the last round it completed -- and the call
 * proceeds with ordinary launches.  Any other call on the handle makes it leave first.  Not used while sc_prover_set_timing is on,
 * on a caller's stream, with sc_prover_set_polling(p, 0) / sc_set_policy("pipeline", 0), or with sc_set_policy("resident", 0).) */
SC_API int sc_prover_set_resident(sc_prover *p, uint32_t patience_polls);
/* MLSumcheck::prove_as_subprotocol pushes the final challenge without binding it (mod.rs:65-67). */
SC_API int sc_prover_push_randomness(sc_prover *p, const uint64_t *r);
/* Copy ProverState back to the host.  randomness: up to num_vars x 4 limbs (may be NULL);
 * tables_out: U x 2^(num_vars - bound) x 4 limbs, bound = max(round-1, 0) (may be NULL);
 * *round, *n_randomness may be NULL. */
SC_API int sc_prover_state(sc_prover *p, uint64_t *randomness, uint32_t *n_randomness, uint64_t *tables_out, uint32_t *round);
SC_API void sc_prover_free(sc_prover *p);
/* Run the handle's kernels on a caller stream (a hipStream_t cast to void*; NULL is the legacy default stream,
 * which is what torch.cuda.current_stream().cuda_stream is unless a side stream is active), or -- use_own != 0 --
 * back on the handle's own non-blocking stream. */
SC_API int sc_prover_set_stream(sc_prover *p, void *hip_stream, int use_own);
/* DEVICE-SIDE WAITS AND FOREIGN HIP TRAFFIC -- the interference contract.  Inside sc_ml_prove* / sc_gkr_prove the latency-bound rounds
 * are launched BEFORE their challenge exists: a one-wavefront kernel (k_wait_challenge) or the persistent tail kernel (k_tail_rounds)
 * sits on the handle's stream and polls a host-mapped word until the calling thread has hashed the previous message.  While such a
 * kernel waits, the calling thread must be able to finish its HIP calls.  The library's own threads are serialised per device (see
 * THREADING); HIP calls made by OTHER code of the process on the same device (hipMalloc / hipFree / synchronous copies take
 * process-wide locks inside the runtime) can hold the calling thread up.  What then happens, in this order:
 *   - nothing, if the delay is shorter than the wait's bound (seconds; policy "wait_spins"): the proof is slower, never different;
 *   - the wait expires: messages computed on a stale challenge are never returned.  A handle whose inputs are intact (borrowed or
 *     streamed tables) proves again from round 0 with synchronous rounds inside the same call; a copying handle returns SC_ERR_HIP
 *     ("... the proof is void") and must be reset with its tables.
 * A host that runs its own HIP work on the device concurrently (a Rust application with its own streams, an ML framework) should
 * switch the device-side waits off for its handles: allow = 0 makes every round of this handle launch after its challenge is known
 * (about 0.3 ms more per 24-variable proof, no kernel ever waits for the host, nothing can expire).  sc_set_policy("pipeline", 0)
 * does the same for every handle of the process.  Returns SC_OK; the setting holds until changed. */
SC_API int sc_prover_set_polling(sc_prover *p, int allow);

/* Sharded use: this handle holds one contiguous high-bit shard of every table.
 * sc_prove_round_partial = prove_round on the shard, result left ON THE DEVICE as (deg+1) x 8 uint64
 * lanes, lane j of evaluation t = 32-bit limb j zero-extended -- summable across ranks with an integer
 * all-reduce (no modular all-reduce exists); asynchronous on the handle's stream.
 * sc_wide_reduce folds summed lanes (host) back to canonical Montgomery limbs. */
SC_API int sc_prove_round_partial(sc_prover *p, const uint64_t *r_or_null, uint64_t *d_wide_out);
SC_API int sc_wide_reduce(const uint64_t *wide, uint32_t n_elems, uint64_t *out);
/* Bind the last local variable: tables of 2 entries -> 1 entry each, written to d_out (U x 4 limbs,
 * device), asynchronous on the handle's stream.  After this the handle is exhausted. */
SC_API int sc_prover_bind_final(sc_prover *p, const uint64_t *r, uint64_t *d_out);

/* Multi-GPU proofs inside the library.  PROCESS MODEL: one sc_prover per GPU, each holding one contiguous
 * high-bit shard of every table (rank g: entries [g * 2^nv_local, (g+1) * 2^nv_local)), driven by one host thread per GPU --
 * either one process per GPU (torchrun, MPI) or one thread per GPU inside one process (sc_set_device is per thread; this is how
 * a Rust host would use it).  There is no single-handle "n_gpus" mode: every rank runs the same call and they meet in the
 * collectives.  A communicator is
 *   - an RCCL one: rank 0 creates the 128-byte unique id (sc_comm_unique_id) and ships it to the other ranks by any means;
 *     every rank then calls sc_comm_init on its own device.  RCCL is bound at run time (dlopen).  The per-round all-reduce
 *     (sum, uint64, (deg+1) x 8 lanes) and the tail's all-gather run on the handle's stream;
 *   - or a HOST transport (sc_comm_init_host): two caller functions that exchange small host buffers (MPI, gloo, shared
 *     memory between threads, ...).  allreduce sums `count` uint64 words in place over all ranks; allgather delivers every
 *     rank's `bytes` bytes in rank order into recv (nranks * bytes).  Both return 0 on success.
 * sc_ml_prove_sharded = MLSumcheck::prove_as_subprotocol (mod.rs:50-70) for the GLOBAL instance of nv_total variables: the
 * nv_total - log2(nranks) local rounds, then bind + all-gather + the last log2(nranks) rounds on the gathered tables.
 * p: this rank's shard, at round 0.  out_proof: nv_total x (deg+1) x 4, out_randomness: nv_total x 4 -- identical on every
 * rank.  rng_or_null: NULL = a fresh transcript.  nranks must be a power of two.  Every rank of a group uses the same kind of handle
 * (all resident or all streamed: the point where the sharding stops depends on it).
 * sc_ml_prove_sharded_rounds: only the first n_rounds local rounds (PolynomialInfo of the global instance is fed first). */
typedef struct sc_comm sc_comm;
typedef int (*sc_allreduce_u64_fn)(void *ctx, uint64_t *inout, size_t count);
typedef int (*sc_allgather_fn)(void *ctx, const void *send, void *recv, size_t bytes);
SC_API int sc_comm_unique_id(uint8_t *out128);
SC_API int sc_comm_init(const uint8_t *id128, int rank, int nranks, sc_comm **out);
SC_API int sc_comm_init_host(int rank, int nranks, sc_allreduce_u64_fn allreduce, sc_allgather_fn allgather, void *ctx, sc_comm **out);
/*   - or PEER TO PEER (sc_comm_init_p2p), for ranks that are THREADS of one process with one GPU each: no collective library at all.
 *     Every rank owns a fine-grained inbox on its device; the ranks meet under `group_id` (any number no other live group uses; the
 *     call blocks until all `nranks` threads have made it -- 60 s at most -- and enables peer access between their devices).  A round's
 *     all-reduce is then ONE small kernel per rank on the prover's stream: it pushes the rank's (deg+1) x 8 lanes into every peer's
 *     inbox as self-validating words (posted writes over xGMI), polls its own inbox, adds, and publishes the total to the host --
 *     instead of ncclAllReduce + a publishing kernel; the tail's gather is peer copies.  Ranks may share a GPU (functional tests on
 *     a one-GPU box; rounds are then not pipelined and the exchange kernel is re-launched until its peers' kernels have run).
 *     LIMITS: at most 16 ranks; round messages of at most 8 evaluations (max_multiplicands <= 7) -- sc_ml_prove_sharded returns
 *     SC_ERR_BAD_ARG for longer ones before it touches the handle.
 *     THREADS AND RCCL: an RCCL communicator driven by one thread per GPU inside one process must follow RCCL's own rule for that
 *     mode -- one communicator per thread, every thread issuing the same collectives in the same order, no thread holding a lock
 *     another needs while inside a collective.  The library's per-device gate is never held across devices, so distinct-device
 *     thread ranks satisfy it; one process per GPU (what bench.py launches) avoids the question, and sc_comm_init_p2p avoids RCCL. */
SC_API int sc_comm_init_p2p(uint64_t group_id, int rank, int nranks, sc_comm **out);
/* diagnostic, collective: one all-reduce and one all-gather of known patterns over the communicator, checked on every rank */
SC_API int sc_comm_selftest(sc_comm *comm);
SC_API void sc_comm_free(sc_comm *comm);
/* what the communicator itself knows: this rank, the number of ranks, and its kind (bench.py reports `ranks_seen` from here, not from
 * its command line) */
#define SC_COMM_RCCL 1
#define SC_COMM_HOST 2
#define SC_COMM_P2P 3
#define SC_COMM_DIRECT_PUBLISH 0x100 /* flag on SC_COMM_RCCL: the communicator's all-reduces deliver a round's (tagged) lanes straight into the
                                      * host-mapped page the host polls -- no publishing kernel behind them (probed by sc_comm_init; sc_set_policy("rccl_direct", 0)
                                      * switches it off) */
SC_API int sc_comm_info(sc_comm *comm, int *rank_out, int *nranks_out, int *kind_out);
/* measurement, collective (every rank calls it): `iters` back-to-back exchanges of n_words (<= 64) uint64 lanes in exactly the form a
 * sharded round uses on this communicator (RCCL: ncclAllReduce into the host-mapped page + the host's poll of the tagged words, or
 * ncclAllReduce + publishing kernel + poll where direct publication is unavailable; peer-to-peer: the one exchange
 * kernel + poll; host transport: publishing kernel + poll + the caller's all-reduce), each waited for before the next is issued; the
 * sums are checked.  *us_mean_out / *us_min_out_or_null: this rank's wall time per exchange.  The per-round cost a sharded proof pays
 * on top of its kernels (the reference's counterpart, the rayon reduce of prover.rs:139-148, is in-process). */
SC_API int sc_comm_exchange_bench(sc_comm *comm, uint32_t n_words, uint32_t iters, double *us_mean_out, double *us_min_out_or_null);
/* how long a host loop waits for a round's message or a peer's lanes before it gives the proof up (default 20 000 ms; also
 * SC_PUBLISH_TIMEOUT_MS in the environment; 0 restores the default).  Process-wide. */
SC_API int sc_set_publish_timeout_ms(uint32_t ms);
SC_API int sc_ml_prove_sharded(sc_prover *p, sc_comm *comm, sc_rng *rng_or_null, uint32_t nv_total, uint64_t *out_proof,
                               uint64_t *out_randomness);
SC_API int sc_ml_prove_sharded_rounds(sc_prover *p, sc_comm *comm, sc_rng *rng, uint32_t nv_total, uint32_t n_rounds, uint64_t *out_proof,
                                      uint64_t *out_randomness);

/* ---- DenseMultilinearExtension::fix_variables (ark-poly; prover.rs:88, gkr mod.rs:122) ------ */
/* in: 2^nv x 4 limbs, point: k x 4 limbs (binds variables 0..k-1, LSB first), out: 2^(nv-k) x 4.
 * flags: SC_TABLES_ON_DEVICE => `in` and `out` are device pointers. */
SC_API int sc_fix_variables(const uint64_t *in, uint32_t nv, const uint64_t *point, uint32_t k, uint64_t *out, uint32_t flags);

/* ---- Blake2b512Rng / FeedableRNG (reference src/rng.rs:11-81) -- host side ------------------- */
SC_API sc_rng *sc_rng_setup(void);
SC_API void sc_rng_free(sc_rng *rng);
SC_API void sc_rng_feed_bytes(sc_rng *rng, const uint8_t *buf, size_t len);             /* feed() after serialization */
SC_API void sc_rng_fill_bytes(sc_rng *rng, uint8_t *dest, size_t len);                  /* RngCore::fill_bytes */
SC_API void sc_rng_feed_poly_info(sc_rng *rng, uint64_t max_multiplicands, uint64_t num_variables); /* feed(&PolynomialInfo) */
SC_API void sc_rng_feed_prover_msg(sc_rng *rng, const uint64_t *evals, uint32_t n);     /* feed(&ProverMsg) */
SC_API void sc_rng_sample_fr(sc_rng *rng, uint64_t *out);                               /* sample_round = F::rand, verifier.rs:128-131 */

/* ---- MLSumcheck::prove / prove_as_subprotocol (reference src/ml_sumcheck/mod.rs:42-70) ------- */
/* rng_or_null: NULL = fresh Blake2b512Rng::setup() (MLSumcheck::prove).  out_proof: num_vars x (deg+1) x 4.
 * out_state_or_null: receives the ProverState handle (caller frees) as prove_as_subprotocol returns it.
 * The latency-bound late rounds do not go back to the host for their launches: rounds with more than 2048 pairs are pipelined (the
 * next round is enqueued behind a wait kernel before the current round's message is hashed), the last rounds (<= 2048 pairs) run in
 * ONE persistent kernel that publishes every message into host-mapped memory and polls a host-mapped mailbox for the challenge;
 * the calling thread only hashes and answers.  Every device-side wait is bounded (seconds); if the calling thread is stalled for
 * longer, messages computed on a stale challenge are never returned: a handle whose inputs are still there (borrowed or streamed
 * tables) proves again from round 0 with synchronous rounds inside the same call; otherwise the call returns SC_ERR_HIP ("... the
 * proof is void") -- reset the handle with its tables and prove again.  sc_set_policy("pipeline", 0) (or a runtime that serialises kernel launches, e.g. a counter-collecting
 * profiler, detected by a probe) turns all of it off: every round is then launched after its challenge is known. */
SC_API int sc_ml_prove(const sc_poly_desc *desc, sc_rng *rng_or_null, uint64_t *out_proof, sc_prover **out_state_or_null);
/* (One-shot use -- MLSumcheck::prove(&poly) in a loop -- does not pay for a prover per call: the library keeps the last one it built
 * here and rewinds it onto the next polynomial of the same structure; with out_state_or_null == NULL device tables are read in
 * place.  sc_release_caches gives the memory back.) */
/* n_rounds x (prove_round, feed, sample) of that loop (mod.rs:57-64) on a handle at round 0, continuing `rng` without feeding
 * PolynomialInfo: the last log2 G rounds of a sharded proof, on the gathered G-entry tables.  out_randomness: n_rounds x 4. */
SC_API int sc_ml_prove_rounds(sc_prover *p, sc_rng *rng, uint32_t n_rounds, uint64_t *out_proof, uint64_t *out_randomness);
/* the same loop on an existing handle at round 0 (sc_prover_init / sc_prover_reset): no allocation per proof */
SC_API int sc_ml_prove_handle(sc_prover *p, sc_rng *rng_or_null, uint64_t *out_proof);

/* ---- verifier (reference src/ml_sumcheck/protocol/verifier.rs:90-251) -- host side, O(nv*deg) - */
/* interpolate_uni_poly (verifier.rs:139-251, including its three tiers: i64 ratios for len <= 20, i128 for len <= 33, field
 * elements beyond).  Every input element must be canonical (< p), else SC_ERR_BAD_ARG. */
SC_API int sc_interpolate_uni_poly(const uint64_t *p_i, uint32_t len, const uint64_t *eval_at, uint64_t *out);
/* MLSumcheck::verify_as_subprotocol (mod.rs:84-100): returns SC_OK / SC_ERR_REJECT;
 * proof: num_vars x (max_multiplicands+1) x 4 limbs; proof_elems: the number of field elements the caller's buffer holds --
 * anything but num_vars * (max_multiplicands+1) is SC_ERR_BAD_ARG "incorrect number of evaluations" (verifier.rs:60-62), so
 * the library never reads past it.  claimed_sum and every proof element must be canonical (< p): the reference's Fp cannot
 * hold anything else, and a non-canonical encoding is rejected with SC_ERR_BAD_ARG before any arithmetic.
 * out_point: num_vars x 4, out_expected: 4 limbs (SubClaim, verifier.rs:29-34). */
SC_API int sc_ml_verify(uint32_t num_vars, uint32_t max_multiplicands, const uint64_t *claimed_sum, const uint64_t *proof,
                 uint64_t proof_elems, sc_rng *rng_or_null, uint64_t *out_point, uint64_t *out_expected);

/* ---- GKR round sumcheck (reference src/gkr_round_sumcheck/mod.rs) ---------------------------- */
/* f1 is a SparseMultilinearExtension over 3*dim variables given as nnz (index, value) pairs with
 * distinct indices (index layout z | x<<dim | y<<2dim, mod.rs:34-35); g / u are host arrays (dim x 4 limbs).
 * flags: SC_TABLES_ON_DEVICE => every table-sized argument (f1_idx, f1_vals, f2, f3 and, for the two initialisation
 * calls, the outputs h_g, f1g_idx, f1g_vals, f1_gu) is a DEVICE pointer: inputs are read in place (never written), outputs
 * are produced in place; without it they are host arrays that the call stages through HBM.  The proof itself, u, v and
 * *f1g_nnz always land on the host.
 * initialize_phase_one (mod.rs:22-42): h_g = 2^dim x 4 out; f1_g out as sorted (index,value) pairs,
 * capacity nnz, *f1g_nnz = count.  A list that arrives in index order (what iterating a BTreeMap gives) is not sorted again;
 * the dense table is accumulated from the list as it came. */
SC_API int sc_gkr_phase_one(const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz, uint32_t dim, const uint64_t *f3,
                     const uint64_t *g, uint32_t flags, uint64_t *h_g, uint64_t *f1g_idx, uint64_t *f1g_vals, uint64_t *f1g_nnz);
/* initialize_phase_two (mod.rs:57-63): f1_gu = 2^dim x 4 out.  Any order of the list; repeated indices add up. */
SC_API int sc_gkr_phase_two(const uint64_t *f1g_idx, const uint64_t *f1g_vals, uint64_t nnz, uint32_t dim, const uint64_t *u,
                     uint32_t flags, uint64_t *f1_gu);
/* GKRRoundSumcheck::prove (mod.rs:93-139).  out_proof: 2 x dim x 3 x 4 limbs (phase1 then phase2
 * messages); out_uv_or_null: 2 x dim x 4 (u then v).
 * The proof needs the two dense tables of the initialisations but not f1(g,.,.) as a list, so this entry point does not sort: the
 * non-zeros' terms are grouped by target cell and added up in LDS (exact field sums in any order give the reference's bits); repeated
 * indices are summed, as a map built by insertion-with-add would; an index-ordered list (BTreeMap order) saves one grouping pass.
 * sc_gkr_phase_one / _two above return the list and take the sorting route. */
SC_API int sc_gkr_prove(sc_rng *rng, const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz, uint32_t dim,
                 const uint64_t *f2, const uint64_t *f3, const uint64_t *g, uint32_t flags, uint64_t *out_proof, uint64_t *out_uv_or_null);
/* f4 -- the same two initialisations with f1's non-zeros SPREAD OVER SEVERAL GPUs (worth it from dim ~ 24).
 * Every rank passes a disjoint subset of f1's (index, value) pairs -- any partition -- and all of f3 / g.  A rank's scatter is a
 * partial sum of a_hg, so the ranks' dense tables are added with ONE table-sized integer all-reduce of the widened limbs
 * (2^dim x 8 uint64 lanes, 64 bytes per entry) and folded back mod p on the device (sc_wide_reduce_table).  f1(g,.,.) stays
 * distributed: f1g_* receive the fold of THIS rank's entries (a key may occur on several ranks; its true value is the sum) and
 * are what the rank passes to sc_gkr_phase_two_sharded.
 *   comm_or_null + h_g (lanes_or_null == NULL): the library all-reduces over `comm` (RCCL or host transport; NULL or one rank =
 *       the unsharded call) and writes the finished table;
 *   lanes_or_null != NULL: no communication -- the rank's contribution is left as lanes for the CALLER's all-reduce (then
 *       sc_wide_reduce_table); h_g_or_null, if given, receives the rank's own partial table.
 * flags: SC_TABLES_ON_DEVICE as for sc_gkr_phase_one (lanes_or_null follows it too). */
SC_API int sc_gkr_phase_one_sharded(sc_comm *comm_or_null, const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz_local, uint32_t dim,
                                    const uint64_t *f3, const uint64_t *g, uint32_t flags, uint64_t *h_g_or_null, uint64_t *lanes_or_null,
                                    uint64_t *f1g_idx, uint64_t *f1g_vals, uint64_t *f1g_nnz);
SC_API int sc_gkr_phase_two_sharded(sc_comm *comm_or_null, const uint64_t *f1g_idx, const uint64_t *f1g_vals, uint64_t nnz_local, uint32_t dim,
                                    const uint64_t *u, uint32_t flags, uint64_t *f1_gu_or_null, uint64_t *lanes_or_null);
/* GKRRoundSumcheck::prove (mod.rs:93-139) over several GPUs END TO END: the two initialisations as above (every rank passes its own
 * subset of f1's non-zeros, and ALL of f2, f3, g) and BOTH sumcheck phases sharded like sc_ml_prove_sharded -- rank r proves over
 * entries [r 2^dim / G, (r+1) 2^dim / G) of the phase's two tables, one all-reduce of the three evaluations per round, early gather,
 * replicated tail; f2(u) is evaluated on every rank and scales the rank's own slice of f3.  `rng` continues the caller's transcript
 * exactly as sc_gkr_prove does (no PolynomialInfo is fed: mod.rs:108-133).  out_proof: 2 x dim x 3 x 4, out_uv_or_null: 2 x dim x 4;
 * identical on every rank.  nranks a power of two below 2^dim.  Worth it from dim ~ 24 (config 5, dim = 20, belongs on one GPU). */
SC_API int sc_gkr_prove_sharded(sc_comm *comm, sc_rng *rng, const uint64_t *f1_idx, const uint64_t *f1_vals, uint64_t nnz_local, uint32_t dim,
                                const uint64_t *f2, const uint64_t *f3, const uint64_t *g, uint32_t flags, uint64_t *out_proof,
                                uint64_t *out_uv_or_null);
/* n elements of 8 summed uint64 lanes -> canonical Montgomery limbs, on the GPU (sc_wide_reduce is the host twin for a handful
 * of elements).  flags: SC_TABLES_ON_DEVICE => lanes / out are device pointers. */
SC_API int sc_wide_reduce_table(const uint64_t *lanes, uint64_t n, uint64_t *out, uint32_t flags);
/* out[i] = scalar * in[i], i < n: `DenseMultilinearExtension::zero() += (f2(u), &f3)` of start_phase2_sumcheck (mod.rs:71-75).
 * flags: SC_TABLES_ON_DEVICE => in / out are device pointers.  scalar: 4 limbs (host). */
SC_API int sc_dense_scale(const uint64_t *in, uint64_t n, const uint64_t *scalar, uint64_t *out, uint32_t flags);

/* ListOfProductsOfPolynomials::evaluate (src/ml_sumcheck/data_structures.rs:99-109): sum_k c_k prod_j T_j(point),
 * the oracle query every reference test ends with (test.rs:71-74) and GKR's f2.evaluate(u) (gkr_round_sumcheck/
 * mod.rs:122).  point: num_vars x 4 limbs, point[0] binds index bit 0.  The U table evaluations run on the device
 * (tables host or device per desc->flags, never modified); out_table_values_or_null receives them (U x 4).
 * num_vars == 0 is allowed here (a table is its single entry). */
SC_API int sc_poly_evaluate(const sc_poly_desc *desc, const uint64_t *point, uint64_t *out_value, uint64_t *out_table_values_or_null);

/* SparseMultilinearExtension::evaluate (ark-poly; the f1(g,u,v) factor of GKRRoundSumcheckSubClaim::verify_subclaim,
 * src/gkr_round_sumcheck/data_structures.rs:33-56): idx distinct, < 2^num_vars, num_vars <= 63; point: num_vars x 4. */
SC_API int sc_sparse_evaluate(const uint64_t *idx, const uint64_t *vals, uint64_t nnz, uint32_t num_vars, const uint64_t *point,
                              uint64_t *out);

/* The GKR entry points keep their device scratch (about 1 GB at dim = 20) and a two-table prover handle in a process-wide
 * cache between calls (allocating and freeing them costs more than a millisecond per call), sc_poly_evaluate / sc_fix_variables keep
 * their work areas (an eighth of the tables) and stream, and sc_ml_prove keeps the last prover it built (bound-table buffers of
 * at most 16 GiB) for the next one-shot proof of the same shape.  This releases all of it. */
SC_API int sc_release_caches(void);
/* Upper bound, in bytes of device memory, of what EACH of those three caches may keep between calls (default 16 GiB; 0 = nothing is
 * kept: every call allocates and frees its own, as a library without caches would).  Lowering the limit releases what is cached now. */
SC_API int sc_set_cache_limit(uint64_t bytes);
/* process-wide counters (monotone; n <= 8 words): which path the latency-bound rounds took and what had to be repeated -- what a host
 * that runs several provers on one GPU cannot see otherwise.  out[0] persistent-tail launches; [1] times a prover found the device's
 * one tail slot taken and ran its late rounds as pipelined launches instead; [2] times the slot was taken over from an interactive
 * handle whose resident kernel had already left; [3] resident kernels started by sc_prove_round; [4] calls that found theirs gone
 * (patience expired) and took the ordinary path; [5] proofs repeated after an expired device-side wait; [6] of the launches in [0],
 * those that kept the tables resident in LDS (k_tail_slices). */
SC_API int sc_library_stats(uint64_t *out, uint32_t n);
/* Library policy: process-wide integers that select between equivalent paths (every path yields the same bits; tests and A/B runs switch
 * them).  The shipped library reads NO environment variable for any of them -- a drop-in library must not change its code path with the
 * caller's environment; it reads only SC_HOST_TRACE / SC_GKR_TRACE (stderr timings) and SC_PUBLISH_TIMEOUT_MS.  Keys (default):
 *   "pipeline" (1)           0: no kernel ever waits for the host (neither the persistent tail kernels nor pipelined launches); per handle:
 *                            SC_NO_DEVICE_POLLING / sc_prover_set_polling
 *   "resident" (1)           0: the interactive sc_prove_round never keeps a kernel on the GPU between calls (per handle: sc_prover_set_resident)
 *   "tail" (1)               0: handles built from now on run their latency-bound rounds as pipelined launches, not in a persistent kernel
 *   "tail_slices" (1)        0: the latency-bound rounds never run out of LDS (k_tail_rounds / launches instead of k_tail_slices)
 *   "vram_mailbox" (1)       0: challenges go through the host-mapped mailbox even where the host could store into device memory
 *   "wide_tree" (1)          0: handles built from now on sum products of five to twelve multiplicands node by node
 *   "rccl_direct" (1)        0: communicators initialised from now on vote against direct publication (every rank then uses the publish kernel)
 *   "shard_gather_log2" (15) 1..15: shard size (log2 entries per table in the first replicated round's region) at which a sharded proof
 *                            gathers onto every rank; must be the same on every rank
 *   "gkr_direct" (1)         0: sc_gkr_prove initialises through sort + merge (the list form) instead of the bucketed kernels
 *   "wait_spins" (2^22)      bound of a device-side wait for a challenge, in polls (tests shorten it to exercise the give-up path)
 *   "staged_init" (1)        0: sc_prover_init over HOST tables copies them whole before round 1 instead of in chunks with round 1 computed
 *                            under the copy (shapes of the merged big-round kernel from 2^18 entries per table)
 * Unknown key or value out of range: SC_ERR_BAD_ARG. */
SC_API int sc_set_policy(const char *key, int64_t value);
SC_API int sc_get_policy(const char *key, int64_t *value);
/* Launch-plan counters (process-wide, monotone): one per path the host side can choose for a round or an initialisation -- which big-round
 * kernel, which table format, which finalize, which latency-bound form, which exchange.  sc_plan_count() plans, sc_plan_name(i) their
 * names ("big.merged.round1", "tail.slices8", "sharded.p2p", ...; NULL past the end), sc_plan_stats fills out[0..n).  What the GPU test
 * suite uses to prove that every plan was reached by a test that compared with the oracle (profiles/plan_coverage.json). */
SC_API uint32_t sc_plan_count(void);
SC_API const char *sc_plan_name(uint32_t i);
SC_API int sc_plan_stats(uint64_t *out, uint32_t n);

/* ---- synthetic inputs + instrumentation (bench / tests) ------------------------------------- */
/* SplitMix64-keyed uniform field elements, generated on the device: n elements of
 * stream `stream` starting at element `first`, written to device memory d_out (n x 4 limbs). */
SC_API int sc_synth_table_device(uint64_t seed, uint64_t stream, uint64_t first, uint64_t n, uint64_t *d_out);
/* Device-side timing of the handle's last sc_prove_round: milliseconds between HIP events recorded
 * on the handle's stream around the round's kernels (excludes the D2H of the evaluations).  The events
 * are only recorded while sc_prover_set_timing(p, 1) is in effect (they cost a few microseconds per round). */
SC_API int sc_prover_last_round_ms(sc_prover *p, float *ms);
/* Per-product instrumentation: with timing on, every product kernel launch is bracketed by HIP events on the
 * handle's stream; sc_prover_get_timing returns the accumulated device milliseconds and launch counts per
 * product (K entries each) and the accumulated per-round span (all kernels of a round incl. finalize) of the rounds
 * that were launched with events: inside sc_ml_prove / sc_gkr_prove the latency-bound late rounds are pipelined
 * (enqueued before their challenge exists) and record none.
 * When a round runs as one launch over all its products (every product has <= 4 multiplicands: k_round_tree),
 * that launch is reported under product 0 and the other products report no launches. */
SC_API int sc_prover_set_timing(sc_prover *p, int on);
SC_API int sc_prover_get_timing(sc_prover *p, double *ms_per_product, uint64_t *launches_per_product, double *rounds_ms);
/* the same per ROUND (num_vars entries, index = round - 1): device time of that round's big-round kernel launch(es) and how many timed
 * proofs contributed; zero for the latency-bound rounds, which record no events.  bench.py's roofline.per_round. */
SC_API int sc_prover_get_round_timing(sc_prover *p, double *ms_per_round, uint64_t *launches_per_round);
/* Rewind a handle to round 0 without reallocating (repeated proofs over resident tables).  Borrowing handle:
 * tables_or_null = new device pointers, or NULL for the same tables.  Copying handle: tables are required and
 * copied in again (device pointers iff flags has SC_TABLES_ON_DEVICE). */
SC_API int sc_prover_reset(sc_prover *p, const uint64_t *const *tables_or_null, uint32_t flags);
/* Host logic of the claim identity, exposed for tests: out[s], s = 0..M, are the weights of a degree-M polynomial's
 * values at the kernel nodes 0, 1, inf (= leading coefficient), -1, 2 in its value at the point r, i.e.
 * f(r) = sum_s out[s] * f(node_s).  1 <= M <= 4; r canonical Montgomery limbs; no device needed. */
SC_API int sc_claim_weights(uint32_t M, const uint64_t *r, uint64_t *out);
/* Elementwise field arithmetic on host arrays of n elements, computed on the GPU (arithmetic parity tests):
 * op 0 mul (production path) | 1 add | 2 sub | 3 mul, plain-C++ CIOS | 4 mul, Comba asm | 5 a[i] * b[0] with b[0] uniform. */
SC_API int sc_fr_elementwise(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n);
/* Elementwise micro-kernel: dependent chains of `reps` field ops per lane, 4 chains per lane (ceilings):
 * variant 0 mul CIOS | 1 add | 2 mul Comba | 3 mul Comba by a uniform operand | 4 two Comba products interleaved. */
SC_API int sc_bench_modmul(uint64_t n_threads, uint32_t reps, uint32_t variant, float *ms_out, uint64_t *checksum_out);

#ifdef __cplusplus
}
#endif
#endif /* SUMCHECK_HIP_H */
