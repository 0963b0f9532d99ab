//! `extern "C"` binding of `libsumcheck_hip.so` (C ABI: `include/sumcheck_hip.h`, SC_ABI_VERSION 5) exposed with the
//! signatures of the reference's public API:
//!
//! | here | reference |
//! |---|---|
//! | [`prover_init`], [`prove_round`]            | `IPForMLSumcheck::{prover_init, prove_round}` `src/ml_sumcheck/protocol/prover.rs:49,74` |
//! | [`prove`], [`prove_as_subprotocol`]         | `MLSumcheck::{prove, prove_as_subprotocol}` `src/ml_sumcheck/mod.rs:42,50` |
//! | [`evaluate`]                                | `ListOfProductsOfPolynomials::evaluate` `src/ml_sumcheck/data_structures.rs:99` |
//! | [`initialize_phase_one`], [`initialize_phase_two`] | `src/gkr_round_sumcheck/mod.rs:22,57` |
//! | [`gkr_prove`]                               | `GKRRoundSumcheck::prove` `src/gkr_round_sumcheck/mod.rs:93` |
//! | [`prove_sharded`]                           | (new) one rank of a multi-GPU `MLSumcheck::prove_as_subprotocol` |
//!
//! The reference crate is `#![forbid(unsafe_code)]` (`src/lib.rs:1`), so the FFI lives in this separate crate.
//! An `Fr` is passed as the address of its 4 x u64 Montgomery limbs: `Fp<MontBackend<FrConfig,4>,4>` is a transparent
//! wrapper of `BigInt<4>([u64; 4])`, which is exactly the C ABI's element layout.
//!
//! `ProverMsg::evaluations` and `GKRProof`'s fields are `pub(crate)` in the reference (`prover.rs:16`,
//! `gkr_round_sumcheck/data_structures.rs:10-11`).  A `ProverMsg` is therefore built here through its derived
//! `CanonicalDeserialize` (a `Vec<F>`: u64 length + elements) -- which keeps `MLSumcheck::verify` and
//! `GKRRoundSumcheck::verify` usable on the result unchanged; a `GKRProof` cannot be built outside the reference crate at all,
//! so [`gkr_prove`] returns the two message lists ([`HipGKRProof`]) and the one-line constructor a maintainer would add
//! to the reference (`GKRProof { phase1_sumcheck_msgs, phase2_sumcheck_msgs }`) is shown in INTEGRATION.md.
//!
//! NOT BUILT in the image this repository was developed in (no cargo); kept in sync with the header by hand.
#![allow(non_camel_case_types)]
use ark_ff::{BigInt, Fp, MontBackend, MontConfig, PrimeField};
use ark_poly::{DenseMultilinearExtension, SparseMultilinearExtension};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use ark_std::os::raw::{c_char, c_int, c_void};
use ark_std::rc::Rc;
use ark_sumcheck::ml_sumcheck::data_structures::ListOfProductsOfPolynomials;
use ark_sumcheck::ml_sumcheck::protocol::prover::{ProverMsg, ProverState};
use ark_sumcheck::ml_sumcheck::protocol::verifier::VerifierMsg;
use ark_sumcheck::ml_sumcheck::protocol::IPForMLSumcheck;
use ark_sumcheck::ml_sumcheck::Proof;
use ark_sumcheck::rng::FeedableRNG;

#[repr(C)]
pub struct sc_poly_desc {
    pub num_vars: u32,
    pub max_multiplicands: u32,
    pub n_products: u32,
    pub coeffs: *const u64,
    pub prod_offsets: *const u32,
    pub prod_indices: *const u32,
    pub n_tables: u32,
    pub tables: *const *const u64,
    pub flags: u32,
}
#[repr(C)]
pub struct sc_prover {
    _private: [u8; 0],
}
#[repr(C)]
pub struct sc_rng {
    _private: [u8; 0],
}
#[repr(C)]
pub struct sc_comm {
    _private: [u8; 0],
}

pub const SC_ABI_VERSION: c_int = 5;
pub const SC_OK: c_int = 0;
pub const SC_ERR_CONSTANT_POLY: c_int = 1;
pub const SC_ERR_FIRST_ROUND_HAS_MSG: c_int = 2;
pub const SC_ERR_MISSING_MSG: c_int = 3;
pub const SC_ERR_NOT_ACTIVE: c_int = 4;
pub const SC_ERR_BAD_ARG: c_int = 5;
pub const SC_TABLES_ON_DEVICE: u32 = 1;
pub const SC_TABLES_BORROW: u32 = 2;
pub const SC_NO_DEVICE_POLLING: u32 = 8; // a host with HIP streams of its own: no kernel of this handle ever waits for the host

pub type sc_allreduce_u64_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, inout: *mut u64, count: usize) -> c_int>;
pub type sc_allgather_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: usize) -> c_int>;

extern "C" {
    pub fn sc_abi_version() -> c_int;
    pub fn sc_last_error() -> *const c_char;
    pub fn sc_set_device(ordinal: c_int) -> c_int;
    pub fn sc_prover_init(desc: *const sc_poly_desc, out: *mut *mut sc_prover) -> c_int;
    pub fn sc_prover_init_streamed(desc: *const sc_poly_desc, chunk_log2: u32, out: *mut *mut sc_prover) -> c_int;
    pub fn sc_prove_round(p: *mut sc_prover, r_or_null: *const u64, out_evals: *mut u64) -> c_int;
    pub fn sc_prover_push_randomness(p: *mut sc_prover, r: *const u64) -> c_int;
    pub fn sc_prover_state(p: *mut sc_prover, randomness: *mut u64, n_randomness: *mut u32, tables_out: *mut u64, round: *mut u32) -> c_int;
    pub fn sc_prover_free(p: *mut sc_prover);
    pub fn sc_release_caches() -> c_int;
    pub fn sc_set_cache_limit(bytes: u64) -> c_int;
    pub fn sc_library_stats(out: *mut u64, n: u32) -> c_int;
    pub fn sc_set_policy(key: *const c_char, value: i64) -> c_int;
    pub fn sc_get_policy(key: *const c_char, value: *mut i64) -> c_int;
    pub fn sc_plan_count() -> u32;
    pub fn sc_plan_name(i: u32) -> *const c_char;
    pub fn sc_plan_stats(out: *mut u64, n: u32) -> c_int;
    pub fn sc_prover_set_polling(p: *mut sc_prover, allow: c_int) -> c_int;
    pub fn sc_prover_set_resident(p: *mut sc_prover, patience_polls: u32) -> c_int;
    pub fn sc_fix_variables(input: *const u64, nv: u32, point: *const u64, k: u32, out: *mut u64, flags: u32) -> c_int;
    pub fn sc_poly_evaluate(desc: *const sc_poly_desc, point: *const u64, out_value: *mut u64, out_table_values_or_null: *mut u64) -> c_int;
    pub fn sc_sparse_evaluate(idx: *const u64, vals: *const u64, nnz: u64, num_vars: u32, point: *const u64, out: *mut u64) -> c_int;
    pub fn sc_ml_prove(desc: *const sc_poly_desc, rng_or_null: *mut sc_rng, out_proof: *mut u64, out_state_or_null: *mut *mut sc_prover) -> c_int;
    pub fn sc_ml_verify(num_vars: u32, max_multiplicands: u32, claimed_sum: *const u64, proof: *const u64, proof_elems: u64,
                        rng_or_null: *mut sc_rng, out_point: *mut u64, out_expected: *mut u64) -> c_int;
    pub fn sc_gkr_phase_one(f1_idx: *const u64, f1_vals: *const u64, nnz: u64, dim: u32, f3: *const u64, g: *const u64, flags: u32,
                            h_g: *mut u64, f1g_idx: *mut u64, f1g_vals: *mut u64, f1g_nnz: *mut u64) -> c_int;
    pub fn sc_gkr_phase_two(f1g_idx: *const u64, f1g_vals: *const u64, nnz: u64, dim: u32, u: *const u64, flags: u32, f1_gu: *mut u64) -> c_int;
    pub fn sc_gkr_prove(rng: *mut sc_rng, f1_idx: *const u64, f1_vals: *const u64, nnz: u64, dim: u32, f2: *const u64, f3: *const u64,
                        g: *const u64, flags: u32, out_proof: *mut u64, out_uv_or_null: *mut u64) -> c_int;
    pub fn sc_gkr_prove_sharded(comm: *mut sc_comm, rng: *mut sc_rng, f1_idx: *const u64, f1_vals: *const u64, nnz_local: u64, dim: u32, f2: *const u64,
                                f3: *const u64, g: *const u64, flags: u32, out_proof: *mut u64, out_uv_or_null: *mut u64) -> c_int;
    pub fn sc_comm_unique_id(out128: *mut u8) -> c_int;
    pub fn sc_comm_init(id128: *const u8, rank: c_int, nranks: c_int, out: *mut *mut sc_comm) -> c_int;
    pub fn sc_comm_init_host(rank: c_int, nranks: c_int, allreduce: sc_allreduce_u64_fn, allgather: sc_allgather_fn, ctx: *mut c_void,
                             out: *mut *mut sc_comm) -> c_int;
    pub fn sc_comm_init_p2p(group_id: u64, rank: c_int, nranks: c_int, out: *mut *mut sc_comm) -> c_int;
    pub fn sc_comm_free(comm: *mut sc_comm);
    pub fn sc_comm_info(comm: *mut sc_comm, rank_out: *mut c_int, nranks_out: *mut c_int, kind_out: *mut c_int) -> c_int;
    pub fn sc_comm_exchange_bench(comm: *mut sc_comm, n_words: u32, iters: u32, us_mean_out: *mut f64, us_min_out_or_null: *mut f64) -> c_int;
    pub fn sc_set_publish_timeout_ms(ms: u32) -> c_int;
    pub fn sc_ml_prove_sharded(p: *mut sc_prover, comm: *mut sc_comm, rng_or_null: *mut sc_rng, nv_total: u32, out_proof: *mut u64,
                               out_randomness: *mut u64) -> c_int;
    pub fn sc_rng_setup() -> *mut sc_rng;
    pub fn sc_rng_free(rng: *mut sc_rng);
    pub fn sc_rng_feed_bytes(rng: *mut sc_rng, buf: *const u8, len: usize);
}

/// Field types whose in-memory form is 4 x u64 Montgomery limbs (BLS12-381 Fr and friends).
pub trait Limbs4: PrimeField {
    fn limbs(&self) -> *const u64;
    fn from_limbs(l: [u64; 4]) -> Self;
    fn to_limbs(&self) -> [u64; 4] {
        unsafe { *(self.limbs() as *const [u64; 4]) }
    }
}
impl<P: MontConfig<4>> Limbs4 for Fp<MontBackend<P, 4>, 4> {
    fn limbs(&self) -> *const u64 {
        self.0 .0.as_ptr()
    }
    fn from_limbs(l: [u64; 4]) -> Self {
        Fp(BigInt(l), core::marker::PhantomData) // raw Montgomery limbs, no conversion
    }
}

fn panic_like_reference(code: c_int) -> ! {
    // the same messages as the reference's panic!s (prover.rs:51,80,91,97)
    match code {
        SC_ERR_CONSTANT_POLY => panic!("Attempt to prove a constant."),
        SC_ERR_FIRST_ROUND_HAS_MSG => panic!("first round should be prover first."),
        SC_ERR_MISSING_MSG => panic!("verifier message is empty"),
        SC_ERR_NOT_ACTIVE => panic!("Prover is not active"),
        _ => {
            let msg = unsafe { std::ffi::CStr::from_ptr(sc_last_error()) }.to_string_lossy().into_owned();
            panic!("libsumcheck_hip: status {code}: {msg}")
        },
    }
}
fn check(rc: c_int) {
    if rc != SC_OK {
        panic_like_reference(rc)
    }
}

/// `ProverMsg { evaluations }` from the library's limbs.  The field is `pub(crate)` in the reference, so the message is
/// rebuilt through its derived `CanonicalDeserialize`: a `Vec<F>` is a u64-LE length followed by the elements.
fn prover_msg<F: Limbs4>(evals: &[[u64; 4]]) -> ProverMsg<F> {
    let v: Vec<F> = evals.iter().map(|l| F::from_limbs(*l)).collect();
    let mut bytes = Vec::new();
    v.serialize_uncompressed(&mut bytes).expect("serialising to a Vec cannot fail");
    ProverMsg::<F>::deserialize_uncompressed_unchecked(&bytes[..]).expect("a serialised Vec<F> is a ProverMsg")
}

struct Flattened {
    coeffs: Vec<[u64; 4]>,
    offsets: Vec<u32>,
    indices: Vec<u32>,
    tables: Vec<*const u64>,
}
impl Flattened {
    fn desc(&self, num_vars: usize, max_multiplicands: usize, flags: u32) -> sc_poly_desc {
        sc_poly_desc {
            num_vars: num_vars as u32,
            max_multiplicands: max_multiplicands as u32,
            n_products: self.coeffs.len() as u32,
            coeffs: self.coeffs.as_ptr() as *const u64,
            prod_offsets: self.offsets.as_ptr(),
            prod_indices: self.indices.as_ptr(),
            n_tables: self.tables.len() as u32,
            tables: self.tables.as_ptr(),
            flags,
        }
    }
}
fn flatten<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>) -> Flattened {
    let coeffs = polynomial.products.iter().map(|(c, _)| c.to_limbs()).collect();
    let mut offsets = vec![0u32];
    let mut indices = Vec::new();
    for (_, idx) in &polynomial.products {
        indices.extend(idx.iter().map(|&i| i as u32));
        offsets.push(indices.len() as u32);
    }
    let tables = polynomial.flattened_ml_extensions.iter().map(|m: &Rc<DenseMultilinearExtension<F>>| m.evaluations.as_ptr() as *const u64).collect();
    Flattened { coeffs, offsets, indices, tables }
}

/// `ProverState` (reference `prover.rs:19-33`) with its tables resident in HBM.  [`HipProverState::to_prover_state`] copies the
/// state back into the reference's own struct (all of whose fields are `pub`).
pub struct HipProverState<F: Limbs4> {
    handle: *mut sc_prover,
    pub list_of_products: Vec<(F, Vec<usize>)>,
    pub n_tables: usize,
    pub num_vars: usize,
    pub max_multiplicands: usize,
}
impl<F: Limbs4> Drop for HipProverState<F> {
    fn drop(&mut self) {
        unsafe { sc_prover_free(self.handle) }
    }
}
impl<F: Limbs4> HipProverState<F> {
    pub fn round(&self) -> usize {
        let mut r = 0u32;
        check(unsafe { sc_prover_state(self.handle, core::ptr::null_mut(), core::ptr::null_mut(), core::ptr::null_mut(), &mut r) });
        r as usize
    }
    /// A host that runs HIP work of its own on the device: `false` = no kernel of this handle ever waits for the host (every round is
    /// launched after its challenge is known); see the interference contract in `sumcheck_hip.h`.
    pub fn set_polling(&mut self, allow: bool) {
        check(unsafe { sc_prover_set_polling(self.handle, allow as c_int) });
    }
    /// Patience (in ~2 us polls; 0 = off) of the kernel that serves the late rounds of `prove_round` called round by round.
    pub fn set_resident(&mut self, patience_polls: u32) {
        check(unsafe { sc_prover_set_resident(self.handle, patience_polls) });
    }
    /// the reference's `ProverState`: randomness, product list, the (partially bound) tables, round
    pub fn to_prover_state(&self) -> ProverState<F> {
        let round = self.round();
        let bound = round.saturating_sub(1);
        let n = 1usize << (self.num_vars - bound);
        let mut rand = vec![[0u64; 4]; self.num_vars + 1];
        let mut n_rand = 0u32;
        let mut tabs = vec![[0u64; 4]; self.n_tables * n];
        check(unsafe { sc_prover_state(self.handle, rand.as_mut_ptr() as *mut u64, &mut n_rand, tabs.as_mut_ptr() as *mut u64, core::ptr::null_mut()) });
        ProverState {
            randomness: rand[..n_rand as usize].iter().map(|l| F::from_limbs(*l)).collect(),
            list_of_products: self.list_of_products.clone(),
            flattened_ml_extensions: tabs
                .chunks(n)
                .map(|t| DenseMultilinearExtension::from_evaluations_vec(self.num_vars - bound, t.iter().map(|l| F::from_limbs(*l)).collect()))
                .collect(),
            num_vars: self.num_vars,
            max_multiplicands: self.max_multiplicands,
            round,
        }
    }
}

/// `IPForMLSumcheck::prover_init` (reference `prover.rs:49-69`): flattens the product list and uploads every unique table once.
pub fn prover_init<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>) -> HipProverState<F> {
    let flat = flatten(polynomial);
    let desc = flat.desc(polynomial.num_variables, polynomial.max_multiplicands, 0);
    let mut handle = core::ptr::null_mut();
    check(unsafe { sc_prover_init(&desc, &mut handle) });
    HipProverState {
        handle,
        list_of_products: polynomial.products.clone(),
        n_tables: polynomial.flattened_ml_extensions.len(),
        num_vars: polynomial.num_variables,
        max_multiplicands: polynomial.max_multiplicands,
    }
}

/// Out-of-core `prover_init` (`sc_prover_init_streamed`): the tables of `polynomial` are NOT copied to HBM as a whole; rounds 1 and 2 stream
/// them from host memory in chunks of `2^chunk_log2` entries (0 = default).  The borrow of `polynomial` makes the compiler hold the
/// tables alive and unchanged for as long as the state exists, which covers the two rounds that read them.
pub fn prover_init_streamed<'a, F: Limbs4>(polynomial: &'a ListOfProductsOfPolynomials<F>, chunk_log2: u32) -> (HipProverState<F>, core::marker::PhantomData<&'a ()>) {
    let flat = flatten(polynomial);
    let desc = flat.desc(polynomial.num_variables, polynomial.max_multiplicands, 0);
    let mut handle = core::ptr::null_mut();
    check(unsafe { sc_prover_init_streamed(&desc, chunk_log2, &mut handle) });
    (
        HipProverState {
            handle,
            list_of_products: polynomial.products.clone(),
            n_tables: polynomial.flattened_ml_extensions.len(),
            num_vars: polynomial.num_variables,
            max_multiplicands: polynomial.max_multiplicands,
        },
        core::marker::PhantomData,
    )
}

/// `IPForMLSumcheck::prove_round` (reference `prover.rs:74-153`), same signature: `&Option<VerifierMsg<F>>` in, `ProverMsg<F>` out.
pub fn prove_round<F: Limbs4>(prover_state: &mut HipProverState<F>, v_msg: &Option<VerifierMsg<F>>) -> ProverMsg<F> {
    let mut out = vec![[0u64; 4]; prover_state.max_multiplicands + 1];
    let r = v_msg.as_ref().map_or(core::ptr::null(), |m| m.randomness.limbs());
    check(unsafe { sc_prove_round(prover_state.handle, r, out.as_mut_ptr() as *mut u64) });
    prover_msg(&out)
}

/// `MLSumcheck::prove_as_subprotocol` (reference `src/ml_sumcheck/mod.rs:50-70`) over the CALLER's transcript: any
/// `FeedableRNG`, fed and sampled here exactly as the reference does, one `sc_prove_round` per round.
pub fn prove_as_subprotocol<F: Limbs4, R: FeedableRNG>(fs_rng: &mut R, polynomial: &ListOfProductsOfPolynomials<F>) -> Result<(Proof<F>, ProverState<F>), R::Error> {
    fs_rng.feed(&polynomial.info())?; // mod.rs:54
    let mut prover_state = prover_init(polynomial);
    let mut verifier_msg = None;
    let mut prover_msgs = Vec::with_capacity(polynomial.num_variables);
    for _ in 0..polynomial.num_variables {
        let prover_msg = prove_round(&mut prover_state, &verifier_msg); // mod.rs:60
        fs_rng.feed(&prover_msg)?;
        prover_msgs.push(prover_msg);
        verifier_msg = Some(IPForMLSumcheck::sample_round(fs_rng)); // mod.rs:63
    }
    let last = verifier_msg.expect("num_variables > 0 (prover_init panics on a constant)").randomness;
    check(unsafe { sc_prover_push_randomness(prover_state.handle, last.limbs()) }); // mod.rs:65-67
    Ok((prover_msgs, prover_state.to_prover_state()))
}

/// `MLSumcheck::prove` (reference `src/ml_sumcheck/mod.rs:42-45`): the whole Fiat-Shamir loop in ONE FFI call (`sc_ml_prove`
/// with a fresh `Blake2b512Rng::setup()` transcript inside the library; the latency-bound rounds run in its persistent kernel).
pub fn prove<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>) -> Proof<F> {
    let flat = flatten(polynomial);
    let desc = flat.desc(polynomial.num_variables, polynomial.max_multiplicands, 0);
    let d = polynomial.max_multiplicands + 1;
    let mut proof = vec![[0u64; 4]; polynomial.num_variables.max(1) * d];
    check(unsafe { sc_ml_prove(&desc, core::ptr::null_mut(), proof.as_mut_ptr() as *mut u64, core::ptr::null_mut()) });
    proof.chunks(d).take(polynomial.num_variables).map(prover_msg).collect()
}

/// The library keeps device memory between calls so that one-shot use costs what a kept prover costs: the last prover it built (up to
/// 16 GiB of bound-table buffers), the work areas of `evaluate` / `fix_variables`, the GKR scratch.  This gives all of it back.
pub fn release_caches() {
    check(unsafe { sc_release_caches() });
}
/// Upper bound (bytes of device memory) of what each of those caches may keep between calls; 0 = nothing is kept.
pub fn set_cache_limit(bytes: u64) {
    check(unsafe { sc_set_cache_limit(bytes) });
}

/// `ListOfProductsOfPolynomials::evaluate` (reference `src/ml_sumcheck/data_structures.rs:99-109`): the oracle query that
/// follows a proof, with every table folded on the GPU (`sc_poly_evaluate`).
pub fn evaluate<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>, point: &[F]) -> F {
    assert_eq!(point.len(), polynomial.num_variables, "wrong number of variables");
    let flat = flatten(polynomial);
    let desc = flat.desc(polynomial.num_variables, polynomial.max_multiplicands, 0);
    let pt: Vec<[u64; 4]> = point.iter().map(|x| x.to_limbs()).collect();
    let mut out = [0u64; 4];
    check(unsafe { sc_poly_evaluate(&desc, pt.as_ptr() as *const u64, out.as_mut_ptr(), core::ptr::null_mut()) });
    F::from_limbs(out)
}

// ---- GKR round sumcheck (reference src/gkr_round_sumcheck/mod.rs) ------------------------------------------------------------
/// `SparseMultilinearExtension.evaluations` (a map index -> value) as the two parallel arrays the C ABI takes
fn sparse_arrays<F: Limbs4>(f: &SparseMultilinearExtension<F>) -> (Vec<u64>, Vec<[u64; 4]>) {
    let mut idx = Vec::with_capacity(f.evaluations.len());
    let mut vals = Vec::with_capacity(f.evaluations.len());
    for (i, v) in f.evaluations.iter() {
        idx.push(*i as u64);
        vals.push(v.to_limbs());
    }
    (idx, vals)
}
fn dense_from_limbs<F: Limbs4>(num_vars: usize, l: &[[u64; 4]]) -> DenseMultilinearExtension<F> {
    DenseMultilinearExtension::from_evaluations_vec(num_vars, l.iter().map(|x| F::from_limbs(*x)).collect())
}

/// `initialize_phase_one` (reference `src/gkr_round_sumcheck/mod.rs:22-42`) -> `(h_g, f1_at_g)`.  The entries of `f1_at_g` are
/// exactly the keys the reference's sparse fold produces, zero-valued ones included.
pub fn initialize_phase_one<F: Limbs4>(f1: &SparseMultilinearExtension<F>, f3: &DenseMultilinearExtension<F>, g: &[F])
                                       -> (DenseMultilinearExtension<F>, SparseMultilinearExtension<F>) {
    let dim = f3.num_vars;
    assert_eq!(f1.num_vars, dim * 3);
    assert_eq!(g.len(), dim);
    let (idx, vals) = sparse_arrays(f1);
    let gl: Vec<[u64; 4]> = g.iter().map(|x| x.to_limbs()).collect();
    let mut h_g = vec![[0u64; 4]; 1 << dim];
    let mut oi = vec![0u64; idx.len().max(1)];
    let mut ov = vec![[0u64; 4]; idx.len().max(1)];
    let mut n1 = 0u64;
    check(unsafe {
        sc_gkr_phase_one(idx.as_ptr(), vals.as_ptr() as *const u64, idx.len() as u64, dim as u32, f3.evaluations.as_ptr() as *const u64,
                         gl.as_ptr() as *const u64, 0, h_g.as_mut_ptr() as *mut u64, oi.as_mut_ptr(), ov.as_mut_ptr() as *mut u64, &mut n1)
    });
    let pairs: Vec<(usize, F)> = (0..n1 as usize).map(|i| (oi[i] as usize, F::from_limbs(ov[i]))).collect();
    (dense_from_limbs(dim, &h_g), SparseMultilinearExtension::from_evaluations(2 * dim, &pairs))
}

/// `initialize_phase_two` (reference `src/gkr_round_sumcheck/mod.rs:57-63`)
pub fn initialize_phase_two<F: Limbs4>(f1_g: &SparseMultilinearExtension<F>, u: &[F]) -> DenseMultilinearExtension<F> {
    assert_eq!(u.len() * 2, f1_g.num_vars);
    let (idx, vals) = sparse_arrays(f1_g);
    let ul: Vec<[u64; 4]> = u.iter().map(|x| x.to_limbs()).collect();
    let mut out = vec![[0u64; 4]; 1 << u.len()];
    check(unsafe { sc_gkr_phase_two(idx.as_ptr(), vals.as_ptr() as *const u64, idx.len() as u64, u.len() as u32, ul.as_ptr() as *const u64, 0, out.as_mut_ptr() as *mut u64) });
    dense_from_limbs(u.len(), &out)
}

/// The two message lists of a `GKRProof` (whose fields are `pub(crate)` in the reference, `data_structures.rs:10-11`).
pub struct HipGKRProof<F: Limbs4> {
    pub phase1_sumcheck_msgs: Vec<ProverMsg<F>>,
    pub phase2_sumcheck_msgs: Vec<ProverMsg<F>>,
    /// the challenges the transcript produced (the verifier's sub-claim point)
    pub u: Vec<F>,
    pub v: Vec<F>,
}

/// The library's own Blake2b512Rng (bit-compatible with `ark_sumcheck::rng::Blake2b512Rng`), for the entry points that run
/// the whole Fiat-Shamir loop inside the library.
pub struct HipRng(*mut sc_rng);
impl HipRng {
    pub fn setup() -> Self {
        HipRng(unsafe { sc_rng_setup() })
    }
    /// `feed(&msg)`: the message's canonical serialisation is absorbed
    pub fn feed<M: CanonicalSerialize>(&mut self, msg: &M) {
        let mut buf = Vec::new();
        msg.serialize_uncompressed(&mut buf).expect("serialising to a Vec cannot fail");
        unsafe { sc_rng_feed_bytes(self.0, buf.as_ptr(), buf.len()) }
    }
}
impl Drop for HipRng {
    fn drop(&mut self) {
        unsafe { sc_rng_free(self.0) }
    }
}

/// `GKRRoundSumcheck::prove` (reference `src/gkr_round_sumcheck/mod.rs:93-139`): sparse fold, scatter, both sumcheck phases and
/// the transcript in ONE FFI call (`sc_gkr_prove`).  `rng` must be in the state the reference's `rng` would be in.
pub fn gkr_prove<F: Limbs4>(rng: &mut HipRng, f1: &SparseMultilinearExtension<F>, f2: &DenseMultilinearExtension<F>, f3: &DenseMultilinearExtension<F>,
                            g: &[F]) -> HipGKRProof<F> {
    assert_eq!(f1.num_vars, 3 * f2.num_vars);
    assert_eq!(f1.num_vars, 3 * f3.num_vars);
    let dim = f2.num_vars;
    assert_eq!(g.len(), dim);
    let (idx, vals) = sparse_arrays(f1);
    let gl: Vec<[u64; 4]> = g.iter().map(|x| x.to_limbs()).collect();
    let mut proof = vec![[0u64; 4]; 2 * dim.max(1) * 3];
    let mut uv = vec![[0u64; 4]; 2 * dim.max(1)];
    check(unsafe {
        sc_gkr_prove(rng.0, idx.as_ptr(), vals.as_ptr() as *const u64, idx.len() as u64, dim as u32, f2.evaluations.as_ptr() as *const u64,
                     f3.evaluations.as_ptr() as *const u64, gl.as_ptr() as *const u64, 0, proof.as_mut_ptr() as *mut u64, uv.as_mut_ptr() as *mut u64)
    });
    let msgs = |off: usize| (0..dim).map(|i| prover_msg(&proof[3 * (off + i)..3 * (off + i) + 3])).collect();
    HipGKRProof {
        phase1_sumcheck_msgs: msgs(0),
        phase2_sumcheck_msgs: msgs(dim),
        u: uv[..dim].iter().map(|l| F::from_limbs(*l)).collect(),
        v: uv[dim..2 * dim].iter().map(|l| F::from_limbs(*l)).collect(),
    }
}

// ---- multi-GPU ------------------------------------------------------------------------------------------------------------------
/// One rank of a multi-GPU `MLSumcheck::prove_as_subprotocol` (`sc_ml_prove_sharded`): call it from one thread per GPU after
/// `sc_set_device(rank)`, with `shard` = this rank's contiguous 1/G slice of every table (`num_variables` = the slice's) and a
/// communicator every rank created from the same unique id (`sc_comm_unique_id` on rank 0, `sc_comm_init` everywhere).
/// Returns the proof and the randomness of the GLOBAL instance, identical on every rank.
pub fn prove_sharded<F: Limbs4>(shard: &ListOfProductsOfPolynomials<F>, comm: *mut sc_comm, nv_total: usize) -> (Proof<F>, Vec<F>) {
    let state = prover_init(shard);
    let d = shard.max_multiplicands + 1;
    let mut proof = vec![[0u64; 4]; nv_total * d];
    let mut rand = vec![[0u64; 4]; nv_total];
    check(unsafe { sc_ml_prove_sharded(state.handle, comm, core::ptr::null_mut(), nv_total as u32, proof.as_mut_ptr() as *mut u64, rand.as_mut_ptr() as *mut u64) });
    (proof.chunks(d).map(prover_msg).collect(), rand.iter().map(|l| F::from_limbs(*l)).collect())
}

#[allow(dead_code)]
fn _unused(_: *mut c_void) {}
