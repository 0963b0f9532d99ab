//! `extern "C"` binding of `libsumcheck_hip.so` (C ABI: `include/sumcheck_hip.h`) exposed with the
//! signatures of `ark_sumcheck::ml_sumcheck::protocol::IPForMLSumcheck::{prover_init, prove_round}`
//! (reference `src/ml_sumcheck/protocol/prover.rs:49,74`) and `MLSumcheck::prove` (`src/ml_sumcheck/mod.rs:42`).
//!
//! The reference crate is `#![forbid(unsafe_code)]` (`src/lib.rs:1`), so the FFI lives in this separate crate.
//! An `Fr` is passed as the address of its 4 x u64 Montgomery limbs: `Fp<MontBackend<FrConfig,4>,4>` is
//! `#[repr(transparent)]`-like over `BigInt<4>([u64; 4])`, which is exactly the C ABI's element layout.
//!
//! NOT BUILT in the image this repository was developed in (no cargo); kept in sync with the header by hand.
#![allow(non_camel_case_types)]
use ark_ff::{BigInt, Fp, MontBackend, PrimeField};
use ark_poly::DenseMultilinearExtension;
use ark_std::os::raw::{c_char, c_int, c_void};
use ark_sumcheck::ml_sumcheck::data_structures::ListOfProductsOfPolynomials;

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

pub const SC_OK: c_int = 0;
pub const SC_ERR_CONSTANT_POLY: c_int = 1;
pub const SC_ERR_FIRST_ROUND_HAS_MSG: c_int = 2;
pub const SC_ERR_MISSING_MSG: c_int = 3;
pub const SC_ERR_NOT_ACTIVE: c_int = 4;

extern "C" {
    pub fn sc_last_error() -> *const c_char;
    pub fn sc_prover_init(desc: *const sc_poly_desc, out: *mut *mut sc_prover) -> c_int;
    pub fn sc_prove_round(p: *mut sc_prover, r_or_null: *const u64, out_evals: *mut u64) -> c_int;
    pub fn sc_prover_push_randomness(p: *mut sc_prover, r: *const u64) -> c_int;
    pub fn sc_prover_state(p: *mut sc_prover, randomness: *mut u64, n_randomness: *mut u32, tables_out: *mut u64, round: *mut u32) -> c_int;
    pub fn sc_prover_free(p: *mut sc_prover);
    pub fn sc_fix_variables(input: *const u64, nv: u32, point: *const u64, k: u32, out: *mut u64, flags: u32) -> c_int;
    pub fn sc_poly_evaluate(desc: *const sc_poly_desc, point: *const u64, out_value: *mut u64, out_table_values_or_null: *mut u64) -> c_int;
    pub fn sc_sparse_evaluate(idx: *const u64, vals: *const u64, nnz: u64, num_vars: u32, point: *const u64, out: *mut u64) -> c_int;
    pub fn sc_ml_prove(desc: *const sc_poly_desc, rng_or_null: *mut sc_rng, out_proof: *mut u64, out_state_or_null: *mut *mut sc_prover) -> c_int;
    pub fn sc_gkr_prove(rng: *mut sc_rng, f1_idx: *const u64, f1_vals: *const u64, nnz: u64, dim: u32, f2: *const u64, f3: *const u64,
                        g: *const u64, out_proof: *mut u64, out_uv_or_null: *mut u64) -> c_int;
    pub fn sc_rng_setup() -> *mut sc_rng;
    pub fn sc_rng_free(rng: *mut sc_rng);
    pub fn sc_rng_feed_bytes(rng: *mut sc_rng, buf: *const u8, len: usize);
}

/// Field types whose in-memory form is 4 x u64 Montgomery limbs (BLS12-381 Fr and friends).
pub trait Limbs4: PrimeField {
    fn limbs(&self) -> *const u64;
    fn from_limbs(l: [u64; 4]) -> Self;
}
impl<P: ark_ff::MontConfig<4>> Limbs4 for Fp<MontBackend<P, 4>, 4> {
    fn limbs(&self) -> *const u64 {
        self.0 .0.as_ptr()
    }
    fn from_limbs(l: [u64; 4]) -> Self {
        Fp(BigInt(l), core::marker::PhantomData) // raw Montgomery limbs, no conversion
    }
}

/// ProverState with its tables resident in HBM (reference `prover.rs:19-33`).
pub struct HipProverState<F: Limbs4> {
    handle: *mut sc_prover,
    pub num_vars: usize,
    pub max_multiplicands: usize,
    _f: core::marker::PhantomData<F>,
}
impl<F: Limbs4> Drop for HipProverState<F> {
    fn drop(&mut self) {
        unsafe { sc_prover_free(self.handle) }
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

/// `IPForMLSumcheck::prover_init` (reference `prover.rs:49-69`): flattens the product list and uploads every
/// unique table once.
pub fn prover_init<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>) -> HipProverState<F> {
    let coeffs: Vec<[u64; 4]> = polynomial.products.iter().map(|(c, _)| unsafe { *(c.limbs() as *const [u64; 4]) }).collect();
    let mut offsets = vec![0u32];
    let mut indices = Vec::new();
    for (_, idx) in &polynomial.products {
        indices.extend(idx.iter().map(|&i| i as u32));
        offsets.push(indices.len() as u32);
    }
    let tables: Vec<*const u64> =
        polynomial.flattened_ml_extensions.iter().map(|m: &std::rc::Rc<DenseMultilinearExtension<F>>| m.evaluations.as_ptr() as *const u64).collect();
    let desc = sc_poly_desc {
        num_vars: polynomial.num_variables as u32,
        max_multiplicands: polynomial.max_multiplicands as u32,
        n_products: polynomial.products.len() as u32,
        coeffs: coeffs.as_ptr() as *const u64,
        prod_offsets: offsets.as_ptr(),
        prod_indices: indices.as_ptr(),
        n_tables: tables.len() as u32,
        tables: tables.as_ptr(),
        flags: 0,
    };
    let mut handle = core::ptr::null_mut();
    let rc = unsafe { sc_prover_init(&desc, &mut handle) };
    if rc != SC_OK {
        panic_like_reference(rc)
    }
    HipProverState { handle, num_vars: polynomial.num_variables, max_multiplicands: polynomial.max_multiplicands, _f: core::marker::PhantomData }
}

/// `IPForMLSumcheck::prove_round` (reference `prover.rs:74-153`): returns `ProverMsg.evaluations`.
pub fn prove_round<F: Limbs4>(state: &mut HipProverState<F>, v_msg: &Option<F>) -> Vec<F> {
    let mut out = vec![[0u64; 4]; state.max_multiplicands + 1];
    let r = v_msg.as_ref().map_or(core::ptr::null(), |r| r.limbs());
    let rc = unsafe { sc_prove_round(state.handle, r, out.as_mut_ptr() as *mut u64) };
    if rc != SC_OK {
        panic_like_reference(rc)
    }
    out.into_iter().map(F::from_limbs).collect()
}

struct Flattened {
    coeffs: Vec<[u64; 4]>,
    offsets: Vec<u32>,
    indices: Vec<u32>,
    tables: Vec<*const u64>,
}
impl Flattened {
    fn desc(&self, num_vars: usize, max_multiplicands: usize) -> sc_poly_desc {
        sc_poly_desc {
            num_vars: num_vars as u32,
            max_multiplicands: max_multiplicands as u32,
            n_products: self.coeffs.len() as u32,
            coeffs: self.coeffs.as_ptr() as *const u64,
            prod_offsets: self.offsets.as_ptr(),
            prod_indices: self.indices.as_ptr(),
            n_tables: self.tables.len() as u32,
            tables: self.tables.as_ptr(),
            flags: 0,
        }
    }
}
fn flatten<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>) -> Flattened {
    let coeffs = polynomial.products.iter().map(|(c, _)| unsafe { *(c.limbs() as *const [u64; 4]) }).collect();
    let mut offsets = vec![0u32];
    let mut indices = Vec::new();
    for (_, idx) in &polynomial.products {
        indices.extend(idx.iter().map(|&i| i as u32));
        offsets.push(indices.len() as u32);
    }
    let tables = polynomial.flattened_ml_extensions.iter().map(|m| m.evaluations.as_ptr() as *const u64).collect();
    Flattened { coeffs, offsets, indices, tables }
}

/// `MLSumcheck::prove` (reference `src/ml_sumcheck/mod.rs:42-45`): the whole Fiat-Shamir loop in one FFI call
/// (`sc_ml_prove` with a fresh `Blake2b512Rng::setup()` transcript inside the library).
pub fn ml_prove<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>) -> Vec<Vec<F>> {
    let flat = flatten(polynomial);
    let desc = flat.desc(polynomial.num_variables, polynomial.max_multiplicands);
    let d = polynomial.max_multiplicands + 1;
    let mut proof = vec![[0u64; 4]; polynomial.num_variables.max(1) * d];
    let rc = unsafe { sc_ml_prove(&desc, core::ptr::null_mut(), proof.as_mut_ptr() as *mut u64, core::ptr::null_mut()) };
    if rc != SC_OK {
        panic_like_reference(rc)
    }
    proof.chunks(d).take(polynomial.num_variables).map(|m| m.iter().map(|l| F::from_limbs(*l)).collect()).collect()
}

/// `ListOfProductsOfPolynomials::evaluate` (reference `src/ml_sumcheck/data_structures.rs:99-109`): the oracle query that
/// follows a proof, with every table folded on the GPU (`sc_poly_evaluate`).
pub fn evaluate<F: Limbs4>(polynomial: &ListOfProductsOfPolynomials<F>, point: &[F]) -> F {
    assert_eq!(point.len(), polynomial.num_variables, "wrong number of variables");
    let flat = flatten(polynomial);
    let desc = flat.desc(polynomial.num_variables, polynomial.max_multiplicands);
    let pt: Vec<[u64; 4]> = point.iter().map(|x| unsafe { *(x.limbs() as *const [u64; 4]) }).collect();
    let mut out = [0u64; 4];
    let rc = unsafe { sc_poly_evaluate(&desc, pt.as_ptr() as *const u64, out.as_mut_ptr(), core::ptr::null_mut()) };
    if rc != SC_OK {
        panic_like_reference(rc)
    }
    F::from_limbs(out)
}

#[allow(dead_code)]
fn _unused(_: *mut c_void) {}
