// kernels.h -- host-callable launchers for the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scd {

constexpr int kBlock = 256;    // 4 wavefronts of 64
constexpr int kMaxFusedM = 8;  // products up to this many multiplicands run register-resident and fused with the bind
constexpr int kMaxGrid = 1024; // 256 CUs x 4 resident 256-thread blocks; longer ranges are grid-strided
constexpr int kMaxSmallTables = 32;         // table-pointer block that fits a kernel argument
constexpr uint64_t kSmallRoundPairs = 1u << 14; // rounds at or below this many pairs are latency-bound: split finer (bind launch + one lane per
                                                // (combination, pair)).  2^16 until round 4: at 2^16 and 2^15 pairs that pair of launches moves 2.2x
                                                // the bytes of the fused tree kernel (every node's combination re-reads its product's tables) and
                                                // is memory-bound, 95 and 58 us a round; swept again with this round's kernels: 2^14 is the minimum
                                                // (config 3 5.125 -> 5.07 ms same-box, profiles/r4l_small_boundary_sweep.txt)

struct FrHost {
    uint64_t l[4];
};

// Evaluation nodes of a product's round polynomial, in the order the kernels use them: 0, 1, inf, -1, 2, -2, 3, ...
// A product with M multiplicands is evaluated at the first M+1 of them.  "inf" is the leading coefficient
// (the product of the slopes), whose operand -- the slope hi-lo -- is needed anyway; -1 and 2 are one lazy
// add away from the table entries.  k_finalize maps the M+1 sums to the message's points 0..deg with a
// host-computed (deg+1) x (M+1) matrix (exact Lagrange weights), so any node set gives the same canonical bits.
constexpr int32_t kNodeInf = 0x7fffffff;
__host__ __device__ constexpr int32_t node_value(int s) {
    return s == 0 ? 0 : s == 1 ? 1 : s == 2 ? kNodeInf : ((s - 3) % 2 == 0 ? -((s - 3) / 2 + 1) : ((s - 3) / 2 + 2));
}

// The round's challenge, prepared by the host for the bind of the tree kernels (fe_device.hpp: fe_mul_bind): row i holds the nine
// 29-bit limbs of (r * 2^(29 i + 58)) mod p as a plain integer, r the challenge's standard (non-Montgomery) value.
struct BindConst {
    int32_t R[9][9];
};

// One distinct table of a product.  mode 0: `src` already holds this round's table (2*n_pairs entries).
// mode 1: `src` holds the previous round's table (4*n_pairs entries); the kernel binds the previous
// challenge on the fly, writes this round's table (2*n_pairs entries) to `dst`, and sums from registers.
struct Slot {
    const uint4 *src;
    uint4 *dst;
    uint32_t mode;
    uint32_t exp; // multiplicity of the table inside the product (e.g. [1,4,4] -> table 4 has exp 2)
    // Internal table format "F29" (big rounds after the first bind): an element is nine signed 29-bit limbs, limbs
    // 0..7 in the 32-byte main array (same stride as the reference layout) and limb 8 in a separate 4-byte array, so
    // a bound table is stored after a carry pass instead of a full canonical reduction + repack, and loads need no
    // unpacking.  src_top == nullptr: `src` is in the reference layout (canonical 8 x 32-bit).  dst is always F29
    // when dst_top != nullptr.
    const int32_t *src_top;
    int32_t *dst_top;
};

struct ProdArgs {
    Slot slot[kMaxFusedM];
    int n_slots;
};
constexpr int kMaxWideM = 12; // products of kMaxFusedM + 1 .. kMaxWideM multiplicands: big rounds as a tree of trees (kernels_wide16.hip)
struct WideArgs16 {
    Slot slot[kMaxWideM]; // one slot per FACTOR, every one in mode 0 (the tables are bound before the launch)
    int n_slots;
};

// One launch for a whole round (every product has <= 4 multiplicands): each block walks all the round's products over its
// own share of the pairs, so a round is one kernel with no launch gaps and the multiplier-bound products (M = 3, 4) of
// some blocks overlap the HBM-bound ones (M = 1, 2) of others.  Partials keep the per-product layout k_finalize reads.
constexpr int kMaxRoundProds = 12;
constexpr int kRoundTreeGrid = 768; // k_round_tree holds 3 blocks per CU (156 VGPRs, 46 KB LDS): one full wave of blocks on 256 CUs
struct TreeProd {
    Slot slot[4];
    uint32_t M;
    uint32_t pad;
    uint64_t partial_off; // in field elements, as in FinProd
};
// the round's finalize step inside k_round_tree (kernels.hip): what k_finalize would have been given
struct RoundFin {
    int enabled;
    int D;
    uint64_t w_off[kMaxRoundProds]; // FinProd::w_off of every product
    const uint4 *Wm;
    uint4 *partials2;               // second-level partials: the layout of the partial array with one "block" per group of 32
    uint32_t *counters;             // device, zero at launch: [0] groups completed, [1 + g] blocks of group g arrived
    uint4 *out;                     // the message: device copy, ...
    uint64_t *out_wide;             // ... widened lanes for an all-reduce (sharded proofs), ...
    uint4 *h_out;                   // ... host-mapped copy with its sequence flag
    uint32_t *h_flag;
    uint32_t seq;
};
struct RoundArgs {
    TreeProd prod[kMaxRoundProds];
    int n_prod;
    RoundFin fin;
    // k_round1_tree_split only: this launch fills blocks [part_block0, part_block0 + gridDim.x) of node rows that are part_stride blocks long
    // (a staged sc_prover_init runs round 1 chunk by chunk under the host-to-device copy, one finalize over all of it); 0 / 0: the whole rows
    uint32_t part_stride, part_block0;
};

// static per-product record for the finalize kernel (device memory)
struct FinProd {
    uint32_t M;           // multiplicands = degree of this product's round polynomial
    uint32_t pad;
    uint64_t partial_off; // offset (in field elements) of this product's partial sums
    uint64_t w_off;       // offset (in field elements) of this product's node->message matrices in d_W:
                          //   [w_off, +D*(M+1))            c_k * W          (partials in the tables' R = 2^256 form)
                          //   [w_off + D*(M+1), +D*(M+1))  c_k * 2^(5(M-1)) * W  (partials from the 2^261-radix kernels, fe_device.hpp)
};

// table pointers passed by value (kernel argument) for the latency-bound small-round kernels
struct TablePtrs {
    const uint4 *src[kMaxSmallTables];
    uint4 *dst[kMaxSmallTables];
    const int32_t *src_top[kMaxSmallTables]; // non-null: that source table is in the internal F29 format
};
// one (product, evaluation point) combination of the small-round sum kernel (device memory, static per prover)
struct Combo {
    uint32_t t;        // evaluation point
    uint32_t M;        // multiplicands of the product
    uint32_t slot_off; // into slot_table / slot_exp
    uint32_t n_slots;
    uint64_t partial_off;
};

// Evaluation at a point (ListOfProductsOfPolynomials::evaluate): every challenge is known up front, so one pass folds up to
// three variables at once -- 8 entries in, 1 out -- and a table moves (1 + 1/8 + ...) x its size instead of 3 x.
constexpr int kFoldMaxLevels = 3;
struct FoldArgs {
    const uint4 *src[kMaxSmallTables];
    uint4 *dst[kMaxSmallTables];
    FrHost r32[kFoldMaxLevels]; // challenges of this pass's variables (LSB first), each times 2^5 (fe_device.hpp radix)
};
// dst[y][i] = fold over `levels` variables of src[y][i << levels ...], i < n_out, y < n_tables (grid.y)
hipError_t launch_fold_multi(const FoldArgs &args, int levels, int n_tables, uint64_t n_out, hipStream_t stream);

constexpr int kMetaProds = 16;
struct FinMeta {
    FinProd prod[kMetaProds];
};
// the small-round sum kernel's metadata as a kernel argument when it fits (no dependent global loads before the table loads)
constexpr int kMetaCombos = 24, kMetaSlots = 48;
struct ComboMeta {
    Combo combo[kMetaCombos];
    uint32_t slot_table[kMetaSlots];
    uint32_t slot_exp[kMetaSlots];
};

// The persistent tail kernel (kernels.hip: k_tail_rounds): every latency-bound round of a proof in one launch.
#ifndef SC_TAIL_MAX_PAIRS // (A/B builds: tools/build_variant.sh NAME -DSC_TAIL_MAX_PAIRS=8192)
#define SC_TAIL_MAX_PAIRS 2048
#endif
constexpr uint64_t kTailMaxPairs = SC_TAIL_MAX_PAIRS; // rounds above this many pairs are throughput- rather than latency-bound: separate (pipelined) launches
                                                      // with full-chip grids beat a resident grid that pays a barrier per phase (measured: 32 vs 54 us at 4096 pairs)
static_assert(kTailMaxPairs >= kBlock && (kTailMaxPairs & (kTailMaxPairs - 1)) == 0 && kTailMaxPairs <= kSmallRoundPairs, "a power of two within the small rounds");
// (at the default, a combination has at most kTailMaxPairs / kBlock = 8 partial blocks: block 0 adds them up with eight lanes each)
constexpr int kTailMaxGrid = 1024; // at most 4 resident blocks per CU (120 VGPRs), all co-resident on a 256-CU device
constexpr int kTailFlatPairs = 16; // rounds with at most this many pairs run in block 0 alone, one lane per (combination, pair)
// ... and with few combinations (one product of two or three multiplicands: the GKR phases, configs 1 and 2) up to 64 pairs do: as long as
// every (combination, pair) has its own lane of the block and a combination's pairs share a wavefront
__host__ __device__ constexpr uint64_t tail_flat_pairs(int n_combos) {
    uint64_t p = kTailFlatPairs;
    while (p < 64 && 2 * p * (uint64_t)n_combos <= (uint64_t)kBlock) p *= 2;
    return p;
}
struct TailTables {
    const uint4 *cur0[kMaxSmallTables];       // where each table's evaluations are when the tail starts ...
    const int32_t *cur0_top[kMaxSmallTables]; // ... non-null: in the internal F29 format (straight from the big rounds)
    uint4 *b0[kMaxSmallTables];               // the 1st, 3rd, ... bind of the tail writes here
    uint4 *b1[kMaxSmallTables];               // the 2nd, 4th, ... here
};
struct TailArgs {
    TailTables t;
    int n_tables;
    int n_rounds;         // rounds this launch runs
    int first_has_bind;   // the first of them binds r0 first (0: it is round 1 of the proof)
    uint64_t first_pairs; // pairs of the first of them (<= kSmallRoundPairs)
    FrHost r0;
    int n_combos, K, D;
    const uint4 *Wm;      // node -> message matrices (FinProd::w_off)
    uint4 *partials;
    uint32_t *sync;       // device, zeroed before the launch: [0] barrier generation [2] challenges released [3] stop [16 + b] arrival flag of block b
    uint4 *sums;          // device, K * D elements: the per-combination sums of a round with many partial blocks
    uint64_t *chal;       // device, 2 x 4 limbs: the challenge of tail round j in slot j & 1
    uint4 *h_out;         // host-mapped: the round message ...
    uint32_t *h_flag;     // ... and its sequence flag (round j publishes seq0 + j)
    uint32_t seq0;
    uint32_t *sig;        // host-mapped: sig[1] = give-up marker of the device-side poll
    const uint64_t *mail_host; // host-mapped, 2 slots x 8 words: the challenge of tail round j, 32-bit limb i as (limb << 32 | sig0 + j) in
                               // word i of slot (sig0 + j) & 1 -- the tag makes every word self-validating, one poll fetches the challenge
    uint32_t sig0;
    uint32_t max_spins;
};
// The same rounds with the tables resident in LDS (kernels_tail.hip: k_tail_slices): block g of B owns a contiguous range of every
// table for the whole launch; one hand-over per round (its sums -> block 0) as self-validating words in `xw`.
constexpr int kTsBlock = 256, kTsMaxBlocks = 256;
constexpr uint64_t kTsMaxPairs = 1u << 16; // it takes over from the first latency-bound round (kSmallRoundPairs) where the slices fit LDS, and up to two
                                           // rounds earlier for shapes with few multiplications per pair (tail_slices_blocks)
// hand-over area (64-bit words): a ring of four accumulator sets (8 groups x kMetaCombos x 8 words) | the last entries of merging blocks
// ([table][block][9 limbs + a filler]) | the challenge as block 0 passes it on
constexpr size_t kTsAccWords = 4 * 8 * (size_t)kMetaCombos * 8;
constexpr size_t kTsXwWords = kTsAccWords + (size_t)kTsMaxBlocks * kMaxSmallTables * 10 + 8;
struct TailSlicesArgs {
    TailArgs base;        // the rounds, tables, finalize data and host mailbox, as k_tail_rounds takes them (sync[3] = the stop word)
    int B;                // blocks: a power of two <= kTsMaxBlocks (tail_slices_blocks)
    uint64_t *xw;         // device, kTsXwWords words, owned by the handle; zero when allocated, its first kTsAccWords zeroed before every launch
    const uint64_t *mail_vram; // non-null: the challenge mailbox in device memory (the host stores into it over the BAR); EVERY block polls it
    uint32_t tag0;        // tag of this launch's first round; >= 1 and above every tag an earlier launch of the handle used
    uint32_t fin_bytes;   // (filled in by the launcher: LDS layout)
    uint32_t lds_entries;
    uint32_t stage_off;
};
// blocks for a tail that starts with `first_pairs` pairs, or 0 when the slices do not fit LDS (the caller then takes k_tail_rounds)
// max_blocks: how many blocks of the kernel the DEVICE holds at once (tail_slices_max_blocks): the blocks wait for each other, so a grid that is
// not co-resident (a CPX / DPX partition, masked CUs) would stall until its waits expire -- fewer, larger slices then, or 0 if those do not fit
int tail_slices_blocks(uint64_t first_pairs, int n_tables, int K, int D, int n_combos, int max_multiplicands, int max_blocks);
int tail_slices_max_blocks(int device, int max_multiplicands); // occupancy of k_tail_slices<.> at its LDS limit x the device's CUs (0: unknown)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, DEVICE): thread ranks of one process drive several GPUs
hipError_t ensure_dynamic_lds(const void *kernel, int bytes, bool (&done)[64]);
hipError_t launch_tail_slices(TailSlicesArgs args, const ComboMeta &meta, const FinMeta &fin, int max_multiplicands, hipStream_t stream);
int tail_max_resident_blocks(int device); // co-resident blocks of the tail kernel (0: unknown -> the tail kernel is not used)
uint32_t wait_spins_default(); // bound of the device-side polls for a challenge (sc_set_policy("wait_spins", n) overrides it: tests)

// ---- library policy: sc_set_policy / sc_get_policy (abi.hip).  Process-wide integers, read where a path is chosen; the shipped library
// reads no environment variable for any of them (the -DSC_EXPERIMENTS build takes their initial values from SC_<KEY> in the environment).
enum PolicyKey {
    kPolPipeline = 0,     // "pipeline"           1: device-side waits allowed (persistent tail kernel, pipelined launches); 0: every round launched after its challenge
    kPolResident,         // "resident"           1: the interactive sc_prove_round may keep a kernel on the GPU between calls
    kPolTailSlices,       // "tail_slices"        1: latency-bound rounds out of LDS (k_tail_slices) where the shape fits; 0: k_tail_rounds / launches
    kPolVramMailbox,      // "vram_mailbox"       1: challenges into device memory over the BAR where the host can store there
    kPolWideTree,         // "wide_tree"          1: products of 5..12 multiplicands as trees (kernels_wide*.hip); 0: node by node
    kPolRcclDirect,       // "rccl_direct"        1: an RCCL round's lanes land in the host-mapped page (if the probe agrees on every rank); 0: publish kernel
    kPolShardGatherLog2,  // "shard_gather_log2"  log2 entries per table and rank region at which a sharded proof gathers (1..15, default 15)
    kPolGkrDirect,        // "gkr_direct"         1: sc_gkr_prove initialises through the bucketed kernels; 0: sort + merge (the list form)
    kPolWaitSpins,        // "wait_spins"         bound of a device-side wait for a challenge, in polls (default 2^22)
    kPolTail,             // "tail"               1: the persistent tail kernels; 0: latency-bound rounds as pipelined launches (what sharded RCCL rounds use)
    kPolStagedInit,       // "staged_init"        1: sc_prover_init over HOST tables copies them in chunks and computes round 1 under the copy (shapes of the merged big-round kernel, >= 2^18 entries)
    kPolCount
};
int64_t policy(int key);

// ---- launch-plan counters: sc_plan_stats / sc_plan_name (abi.hip).  One counter per path the host side can choose for a round (or an
// initialisation); bumped where the choice is made.  tests/conftest.py accumulates them over the GPU suite per test and
// tests/test_zz_plan_coverage.py asserts that every plan was reached by a test that also computed the oracle's answer.
enum Plan {
    kPlanBigMergedRound1 = 0, // k_round1_tree_split: all products, one launch, no bind
    kPlanBigMergedBindChain,  // k_round_tree_split<chain>: round 2 (sources canonical)
    kPlanBigMergedBind,       // k_round_tree_split: rounds >= 3 (sources F29 or canonical)
    kPlanBigClaimIdentity,    // ... with node 1 from the claim identity (kSkip1)
    kPlanBigF29Store,         // a big binding round stored its tables in the internal F29 format
    kPlanBigCanonicalStore,   // ... in the reference layout (lists that keep canonical tables)
    kPlanBigPerProductTree,   // k_prod_tree<M <= 4>: one launch per product (more products than a merged launch takes)
    kPlanBigWide,             // k_prod_tree_wide<5..8>
    kPlanBigWide16,           // k_prod_tree_wide16<9..12>
    kPlanBigGeneric,          // k_sum_generic (13 and more, or wide_tree = 0 beyond 8)
    kPlanBigNodeByNode,       // k_prod_round_fe (wide_tree = 0, 5..8)
    kPlanBigBindPass,         // k_fix_multi up front (lists with products beyond kMaxFusedM)
    kPlanBigStreamed,         // rounds 1-2 of a streamed handle, chunk by chunk
    kPlanBigStagedRound1,     // round 1 computed inside sc_prover_init / sc_prover_reset, chunk by chunk under the host-to-device copy
    kPlanFinalizeMultiBlock,  // k_finalize_mb
    kPlanFinalizeOneBlock,    // k_finalize (more products than a launch's arguments describe: metadata from device memory)
    kPlanFinalizeNoLds,       // k_finalize without LDS staging (node sums beyond 48 KB)
    kPlanSmallLaunched,       // k_fix_multi + k_sum_combos_meta after the challenge
    kPlanSmallCombosTable,    // ... k_sum_combos (combination metadata from device memory)
    kPlanSmallPtrs,           // ... k_sum_combos_ptrs (more than 32 tables)
    kPlanSmallPipelined,      // a latency-bound round enqueued behind k_wait_challenge
    kPlanTailSlices8,         // k_tail_slices<8>
    kPlanTailSlices12,        // k_tail_slices<12>
    kPlanTailRounds,          // k_tail_rounds
    kPlanResidentSlices,      // the interactive protocol's resident kernel: k_tail_slices
    kPlanResidentRounds,      // ... k_tail_rounds
    kPlanShardedRcclDirect,   // sharded rounds: ncclAllReduce into the host-mapped page
    kPlanShardedRcclPublish,  // ... all-reduce + publish kernel
    kPlanShardedHost,         // ... the caller's host transport
    kPlanShardedP2P,          // ... k_p2p_allreduce
    kPlanShardedGatherTail,   // bind + all-gather + replicated tail
    kPlanGkrBucketedGrouped,  // GKR initialisation: buckets of an index-ordered list (k_bucket_bounds)
    kPlanGkrBucketedCounted,  // ... counting sort into buckets
    kPlanGkrListForm,         // ... sort + merge + scatter
    kPlanGkrCoeffFromBound,   // phase two's coefficient f2(u) from phase one's bound table
    kPlanGkrSharded,          // sc_gkr_prove_sharded
    kPlanFoldMulti,           // sc_poly_evaluate / sc_fix_variables (k_fold_multi)
    kPlanCount
};
void plan_hit(int plan);
hipError_t launch_scale_w_by_table_eval(FrHost *W, const FrHost *W0, uint32_t n, const void *table, const FrHost &r, hipStream_t stream); // W = W0 * table(r)
hipError_t launch_zero_words(uint32_t *p, uint32_t n, hipStream_t stream, uint32_t *p2 = nullptr, uint32_t n2 = 0); // (a plain kernel: hipMemsetAsync may take runtime paths that wait on other streams)
hipError_t launch_tail_rounds(const TailArgs &args, const ComboMeta &meta, const FinMeta &fin, int grid, hipStream_t stream);

int grid_for_pairs(uint64_t n_pairs);

// product k of one round: partials[t*grid+blk] = sum over this block's pairs of prod_j line_j(t), t = 0..M (node-major, so the
// finalize kernel reads each node's partials as one contiguous run)
#ifdef SC_EXPERIMENTS // saturated-arithmetic cross-check kernel
hipError_t launch_prod_round(int M, const ProdArgs &args, const FrHost &r, uint64_t n_pairs, FrHost *d_partials, int grid,
                             hipStream_t stream);
#endif
// the same in carry-free 29-bit-limb arithmetic; r32 = challenge * 2^5, partials carry 2^(-5(M-1))
hipError_t launch_prod_round_fe(int M, const ProdArgs &args, const FrHost &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                                hipStream_t stream);
// production big-round kernel: args list exactly M factors (repeated tables listed repeatedly; modes 0 / 1 / 3), static
// multiplication tree per M; same partial layout and 2^(-5(M-1)) scaling as launch_prod_round_fe
hipError_t launch_prod_tree(int M, const ProdArgs &args, const BindConst &rc, uint64_t n_pairs, FrHost *d_partials, int grid,
                            hipStream_t stream);
// five to eight factors: the halves' product trees, extended to the product's nodes by integer combinations, one product per node
// (kernels_wide.hip); launch_prod_tree forwards to it
hipError_t launch_prod_tree_wide(int M, const ProdArgs &args, const BindConst &rc, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream);
// nine to kMaxWideM multiplicands over bound tables; comp = 2^(5(M-1)) in Montgomery form: the sums leave in k_sum_generic's form
hipError_t launch_prod_tree_wide16(int M, const WideArgs16 &args, const FrHost &comp, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream);
// all products of a round in one launch (see RoundArgs); d_partials is the base of the partial-sum array
// one product per block row: grid x n_prod blocks, `grid` partial blocks per product.  (split = false, experiments build only: the
// previous kernels, every block walking all products)
// skip1: node 1 of every product is left out (binding rounds only); the round's finalize launch must then carry ClaimArgs
hipError_t launch_round_tree(const RoundArgs &args, const BindConst &rc, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream, bool split = true,
                             bool skip1 = false);
#ifdef SC_EXPERIMENTS // tiled variant (LDS-staged, one wavefront per node): grid from grid_for_tiles, same partial layout and scaling as _fe
int grid_for_tiles(uint64_t n_pairs);
hipError_t launch_round_tile(int M, const ProdArgs &args, const FrHost &r32, uint64_t n_pairs, FrHost *d_partials, int grid,
                             hipStream_t stream);
#endif
// generic (any M): tables already bound; slot lists in device memory
hipError_t launch_sum_generic(const uint4 *const *d_cur_tables, const uint32_t *d_slot_table, const uint32_t *d_slot_exp,
                              int n_slots, int M, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream);
// out[b] = in[2b] + r*(in[2b+1]-in[2b]), b < n_out
hipError_t launch_fix(const uint4 *src, uint4 *dst, const FrHost &r, uint64_t n_out, hipStream_t stream);
// small rounds: bind every table in one launch (grid.y = table) ...
// one-wave kernel that holds the stream until flag_dev[0] == want (host-mapped word; bounded spin -- on giving up it stores want
// to flag_dev[1]), then copies the challenge from the host-mapped mailbox to device memory
hipError_t launch_wait_challenge(uint32_t *flag_dev, uint32_t want, const FrHost *mail_host_dev, FrHost *mail_dev, hipStream_t stream,
                                 uint32_t spins_override = 0);
// r_mail (device-visible, may be host-mapped) overrides r when non-null: the challenge is fetched at run time
hipError_t launch_fix_multi(const TablePtrs &tp, int n_tables, const FrHost &r, const FrHost *r_mail, uint64_t n_out, hipStream_t stream);
// ... and one launch for every (product, evaluation point) combination (grid.y = combination), one lane per pair
hipError_t launch_sum_combos(const TablePtrs &tp, const Combo *d_combos, int n_combos, const uint32_t *d_slot_table,
                             const uint32_t *d_slot_exp, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream);
// the same with the metadata passed by value (n_combos <= kMetaCombos, slot entries <= kMetaSlots)
// the same for more tables than TablePtrs holds: pointers from a device array (table index -> this round's evaluations)
hipError_t launch_sum_combos_ptrs(const uint4 *const *d_cur_tables, const Combo *d_combos, int n_combos, const uint32_t *d_slot_table,
                                  const uint32_t *d_slot_exp, uint64_t n_pairs, FrHost *d_partials, int grid, hipStream_t stream);
hipError_t launch_sum_combos_meta(const TablePtrs &tp, const ComboMeta &meta, int n_combos, uint64_t n_pairs, FrHost *d_partials, int grid,
                                  hipStream_t stream);
// combine per-block partials of all products into the round polynomial (D evaluations)
// Node 1 from the previous round.  For every product, S(0) + S(1) of a binding round equals the previous round's polynomial of that
// product at the challenge just bound (the verifier's check, product by product), so a big round that follows one whose complete node
// sums are still on the device computes the nodes {0, inf, -1, 2} only: one final Montgomery product less per product and pair.
// The finalize launch gets the previous sums and the weights lam_s(r) of the node basis at the challenge (host, once per round, while
// the round kernel runs) and restores S(1) = sum_s lam_s S_prev(s) - S(0) before the message is formed.  Sums of both rounds carry the
// same 2^(-5(M-1)), so the identity holds between them as stored.
struct ClaimArgs {
    uint32_t skip1;    // 1: the round kernel left node 1 out
    uint32_t pad;
    const uint4 *prev; // the previous round's K * D node sums (the multi-block finalize's `d_scratch` of that round)
    FrHost lam[14];    // lam[claim_off(M) + s], s = 0..M, for M = 1..4
};
__host__ __device__ constexpr int claim_off(int M) { return (M - 1) * (M + 2) / 2; }
// Widened lanes (8 x 64-bit per evaluation: 32-bit limbs zero-extended, summable across ranks) may carry a TAG in their top bits:
// every rank's finalize step ORs wide_tag_of(seq) << kWideTagShift into its lanes, the integer all-reduce adds the tags up, and a word
// whose top bits read nranks * tag is this round's total -- RCCL can then deliver straight into host-mapped memory and the host's
// poll is the fetch (no publish kernel, no flag).  The lanes themselves stay below 2^40 (sums of 32-bit limbs over <= 16 ranks).
constexpr int kWideTagShift = 44;
__host__ __device__ constexpr uint32_t wide_tag_of(uint32_t seq) { return (seq & 0x7fffu) + 1u; } // 1 .. 2^15: nranks * tag < 2^20 for <= 16 ranks
bool finalize_keeps_sums(int K, int D, int nblocks, bool have_host_prods, bool have_counter);
// h_prods_or_null: host copy of the records; with at most kMetaProds products they travel as a kernel argument
hipError_t launch_finalize(const FinProd *d_prods, const FinProd *h_prods_or_null, const FrHost *d_W, int K, int D, int nblocks, const FrHost *d_partials, FrHost *d_scratch,
                           FrHost *d_out, uint64_t *d_out_wide, FrHost *h_out_mapped, uint32_t *h_flag_mapped, uint32_t seq,
                           int scaled /* bit 0: scaled partials; bit 1: tagged lanes */, uint32_t *d_counter_or_null, hipStream_t stream, const ClaimArgs *claim_or_null = nullptr);
// (d_counter_or_null: one zeroed device word owned by the caller selects the multi-block form, which needs d_scratch for K * D sums
// and leaves the word at zero)
hipError_t launch_synth(uint64_t seed, uint64_t stream_id, uint64_t first, uint64_t n, uint4 *d_out, hipStream_t stream);
hipError_t launch_scale(const uint4 *src, uint4 *dst, const FrHost &s, uint64_t n, hipStream_t stream);
hipError_t launch_tag_words(uint64_t *d_words, int n, uint32_t gen, hipStream_t stream); // words |= wide_tag_of(gen) << kWideTagShift
hipError_t launch_publish_words(const uint64_t *d_src, uint64_t *h_dst_mapped, int n, uint32_t *h_flag_mapped, uint32_t seq, hipStream_t stream);
// recv[g][u][e] -> tabs[u][g * per + e] (elements of 32 bytes): the all-gathered remainders of G shards become U tables of G * per entries
// Peer-to-peer all-reduce of a round's lanes (sc_comm_init_p2p): rank r PUSHES its n_words lanes into every peer's inbox (posted
// writes over xGMI, or plain device memory when ranks share a GPU) as self-validating words (generation << 40 | value: the lanes are
// sums of 32-bit limbs, far below 2^40), then polls its OWN inbox -- local memory -- until every source's words carry the generation,
// adds them up, leaves the totals in `lanes` and publishes them to the host-mapped page like k_publish_words.  No collective library,
// no second launch, no flag/data ordering to get wrong: a word that shows the generation IS the data.
// inbox layout (uint64 words): [generation & 1][source rank][kP2PWords].
constexpr int kP2PMaxRanks = 16, kP2PWords = 64;
constexpr size_t kP2PInboxWords = 2 * (size_t)kP2PMaxRanks * kP2PWords;
constexpr uint32_t kP2PRetryBit = 0x80000000u; // h_flag = seq | kP2PRetryBit: a peer had not arrived within max_spins (the host launches the kernel again)
struct P2PArgs {
    uint64_t *inbox[kP2PMaxRanks]; // every rank's inbox, addressable from this device
    int nranks, rank, n_words;
    uint32_t gen;                  // generation of this exchange, 1, 2, ... (24 bits are compared)
    uint32_t max_spins;
};
struct PeerLanes {
    const uint64_t *p[kP2PMaxRanks];
    int n;
};
hipError_t launch_sum_peer_lanes(const PeerLanes &peers, uint64_t n_words, uint64_t *out, hipStream_t stream); // out[i] = sum_q peers.p[q][i]
hipError_t launch_p2p_allreduce(const P2PArgs &args, uint64_t *d_lanes, uint64_t *h_dst_mapped, uint32_t *h_flag_mapped, uint32_t seq, hipStream_t stream);
hipError_t launch_gather_to_tables(const uint4 *recv, uint4 *tabs, uint32_t G, uint32_t U, uint32_t per, hipStream_t stream);
// acc (+)= in (D elements; first: acc = in); last: the sum also goes to d_out / the host-mapped page with its sequence flag (D <= 64)
hipError_t launch_msg_accumulate(const FrHost *in, FrHost *acc, int D, bool first, bool last, FrHost *d_out, uint64_t *d_out_wide, FrHost *h_out_mapped,
                                 uint32_t *h_flag_mapped, uint32_t seq, hipStream_t stream);
// F29 table -> canonical reference layout (state export)
hipError_t launch_f29_to_sat(const uint4 *src, const int32_t *src_top, uint4 *dst, uint64_t n, hipStream_t stream);
hipError_t launch_fr_elementwise(int op, const uint4 *a, const uint4 *b, const FrHost &u, uint4 *out, uint64_t n, hipStream_t stream);
hipError_t launch_bench_modmul(uint64_t n_threads, uint32_t reps, uint32_t variant, uint64_t *d_sink, hipStream_t stream);

} // namespace scd
