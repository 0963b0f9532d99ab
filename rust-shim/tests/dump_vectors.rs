//! THE PIN: runs REAL arkworks (ark-sumcheck + the algebra git master it patches to) on the inputs of tests/golden/*.json and
//! compares, byte for byte, everything this repository restated from memory instead of from source:
//!
//!   * `F::rand` over a `Blake2b512Rng`, `RngCore::fill_bytes`, `feed` (transcript.json: squeeze vectors and field samples);
//!   * `CanonicalSerialize` of `PolynomialInfo`, `ProverMsg`, `Proof` (ml_*.json: "fs_proof" as the serialised proof bytes);
//!   * the challenges of a whole non-interactive proof (ml_*.json: "fs_randomness" vs `ProverState::randomness`) and the verifier's
//!     sub-claim ("subclaim_expected");
//!   * THE HOT PATH ITSELF, round by round with fixed (non-transcript) challenges: `IPForMLSumcheck::prover_init / prove_round`
//!     (ml_*.json: "challenges" -> "rounds", and the bound tables left in `ProverState` -> "final_tables"), and the Montgomery limbs
//!     of an element as they sit in memory ("tables_mont0": what crosses the C ABI);
//!   * `SparseMultilinearExtension::fix_variables` inside `initialize_phase_one / _two` (gkr_*.json: "h_g", "f1_g_idx",
//!     "f1_g_vals" -- zero-valued entries included -- and "f1_gu");
//!   * `GKRRoundSumcheck::prove` + `verify` end to end (gkr_*.json: "sum", "u", "v", "expected": the sub-claim depends on every
//!     message of both phases through the transcript), and the messages of both phases themselves ("phase1", "phase2": crate-private in
//!     `GKRProof`, so the test replays `GKRRoundSumcheck::prove`'s loop from its public building blocks).
//! tests/test_oracle.py::test_pin_reads_every_fixture_field fails when a fixture gains a field this file does not read.
//!
//! One command on a machine with cargo:  `cd rust-shim && cargo test --release --test dump_vectors -- --nocapture`
//! (no GPU and no libsumcheck_hip needed: set SUMCHECK_HIP_LIB_DIR to any directory holding a stub if the linker insists).
//! Every comparison names the first differing field, so a failure says WHICH recalled semantic is wrong.
//! NOT BUILT in this repository's image (no cargo).
use ark_ff::{BigInteger, PrimeField};
use ark_poly::{DenseMultilinearExtension, Polynomial, SparseMultilinearExtension};
use ark_serialize::{CanonicalSerialize, Compress, SerializationError, Valid};
use ark_std::io::Write;
use ark_std::rand::RngCore;
use ark_std::rc::Rc;
use ark_std::UniformRand;
use ark_sumcheck::gkr_round_sumcheck::{initialize_phase_one, initialize_phase_two, start_phase1_sumcheck, start_phase2_sumcheck, GKRRoundSumcheck};
use ark_sumcheck::ml_sumcheck::data_structures::{ListOfProductsOfPolynomials, PolynomialInfo};
use ark_sumcheck::ml_sumcheck::protocol::verifier::VerifierMsg;
use ark_sumcheck::ml_sumcheck::protocol::IPForMLSumcheck;
use ark_sumcheck::ml_sumcheck::MLSumcheck;
use ark_sumcheck::rng::{Blake2b512Rng, FeedableRNG};
use ark_test_curves::bls12_381::Fr;
use serde_json::Value;

fn unhex(s: &str) -> Vec<u8> {
    (0..s.len()).step_by(2).map(|i| u8::from_str_radix(&s[i..i + 2], 16).unwrap()).collect()
}
fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{x:02x}")).collect()
}
/// fixtures store canonical integers as 64 big-endian hex digits
fn fr(v: &Value) -> Fr {
    Fr::from_be_bytes_mod_order(&unhex(v.as_str().unwrap()))
}
fn fr_hex(x: &Fr) -> String {
    hex(&x.into_bigint().to_bytes_be())
}
fn frs(v: &Value) -> Vec<Fr> {
    v.as_array().unwrap().iter().map(fr).collect()
}
fn load(name: &str) -> Value {
    serde_json::from_str(&std::fs::read_to_string(format!("../tests/golden/{name}.json")).unwrap()).unwrap()
}
/// 32 bytes, little-endian, of the canonical integer: what `CanonicalSerialize for Fp` is believed to emit
fn fr_le32(x: &Fr) -> Vec<u8> {
    x.into_bigint().to_bytes_le()
}
fn first_diff(what: &str, got: &[u8], want: &[u8]) {
    assert_eq!(got.len(), want.len(), "{what}: length {} != expected {}", got.len(), want.len());
    if let Some(i) = (0..got.len()).find(|&i| got[i] != want[i]) {
        panic!("{what}: first difference at byte {i}: got {:02x}, expected {:02x}", got[i], want[i]);
    }
}

/// raw bytes fed as they are (the fixtures' "feed" ops; the reference feeds `b"..."` arrays, which serialise without a length)
struct Raw(Vec<u8>);
impl Valid for Raw {
    fn check(&self) -> Result<(), SerializationError> {
        Ok(())
    }
}
impl CanonicalSerialize for Raw {
    fn serialize_with_mode<W: Write>(&self, mut w: W, _c: Compress) -> Result<(), SerializationError> {
        w.write_all(&self.0)?;
        Ok(())
    }
    fn serialized_size(&self, _c: Compress) -> usize {
        self.0.len()
    }
}

#[test]
fn transcript_vectors() {
    let t = load("transcript");
    let mut rng = Blake2b512Rng::setup();
    for (n, op) in t["ops"].as_array().unwrap().iter().enumerate() {
        match op[0].as_str().unwrap() {
            "feed" => rng.feed(&Raw(unhex(op[1].as_str().unwrap()))).unwrap(),
            "fill" => {
                let mut buf = vec![0u8; op[1].as_u64().unwrap() as usize];
                rng.fill_bytes(&mut buf);
                first_diff(&format!("transcript op {n} (fill_bytes {})", buf.len()), &buf, &unhex(op[2].as_str().unwrap()));
            },
            _ => assert_eq!(fr_hex(&Fr::rand(&mut rng)), op[1].as_str().unwrap(), "transcript op {n}: F::rand"),
        }
    }
    // RFC 7693's "abc": the first 64 bytes squeezed after feeding b"abc" ARE the BLAKE2b-512 digest of "abc" (fill_bytes finalises a clone)
    let mut rng = Blake2b512Rng::setup();
    rng.feed(&Raw(b"abc".to_vec())).unwrap();
    let mut abc = [0u8; 64];
    rng.fill_bytes(&mut abc);
    first_diff("blake2b-512(\"abc\") through feed + fill_bytes", &abc, &unhex(t["blake2b_abc"].as_str().unwrap()));
    // feed(&PolynomialInfo), feed(&ProverMsg)-shaped Vec<F>, then F::rand and 64 more bytes
    let s = &t["structured"];
    let mut rng = Blake2b512Rng::setup();
    rng.feed(&PolynomialInfo { max_multiplicands: s["info"][0].as_u64().unwrap() as usize, num_variables: s["info"][1].as_u64().unwrap() as usize }).unwrap();
    rng.feed(&frs(&s["msg"])).unwrap(); // ProverMsg { evaluations: Vec<F> } serialises as its Vec<F>
    assert_eq!(fr_hex(&Fr::rand(&mut rng)), s["sample"].as_str().unwrap(), "structured: F::rand after feed(info), feed(msg)");
    let mut buf = [0u8; 64];
    rng.fill_bytes(&mut buf);
    first_diff("structured: next 64 bytes", &buf, &unhex(s["next64"].as_str().unwrap()));
}

fn build_poly(case: &Value) -> ListOfProductsOfPolynomials<Fr> {
    let nv = case["nv"].as_u64().unwrap() as usize;
    let tables: Vec<Rc<DenseMultilinearExtension<Fr>>> =
        case["tables"].as_array().unwrap().iter().map(|t| Rc::new(DenseMultilinearExtension::from_evaluations_vec(nv, frs(t)))).collect();
    let mut poly = ListOfProductsOfPolynomials::new(nv);
    for (k, shape) in case["shapes"].as_array().unwrap().iter().enumerate() {
        let c = fr(&case["products"][k][0]);
        poly.add_product(shape.as_array().unwrap().iter().map(|i| tables[i.as_u64().unwrap() as usize].clone()), c);
    }
    poly
}

#[test]
fn ml_proof_vectors() {
    for name in ["nv1_trivial", "nv2_single", "nv3_c1shape", "nv5_deg12", "nv6_c3shape", "nv6_shared", "nv7_c2shape", "nv8_bench"] {
        let case = load(&format!("ml_{name}"));
        let poly = build_poly(&case);
        assert_eq!(poly.flattened_ml_extensions.len(), case["flattened_table_ids"].as_array().unwrap().len(), "{name}: de-duplicated tables");
        // the whole non-interactive proof, as bytes: Vec<ProverMsg> = u64 count, then per message u64 count + 32-byte elements
        let (proof, state) = MLSumcheck::prove_as_subprotocol(&mut Blake2b512Rng::setup(), &poly).unwrap();
        let mut got = Vec::new();
        proof.serialize_uncompressed(&mut got).unwrap();
        let rounds = case["fs_proof"].as_array().unwrap();
        let mut want = (rounds.len() as u64).to_le_bytes().to_vec();
        for r in rounds {
            let ev = frs(r);
            want.extend_from_slice(&(ev.len() as u64).to_le_bytes());
            for e in &ev {
                want.extend_from_slice(&fr_le32(e));
            }
        }
        assert_eq!(got.len(), want.len(), "{name}: serialised proof length (CanonicalSerialize layout)");
        let d = case["fs_proof"][0].as_array().unwrap().len();
        if let Some(i) = (0..got.len()).find(|&i| got[i] != want[i]) {
            let per_msg = 8 + 32 * d;
            let (round, off) = ((i.saturating_sub(8)) / per_msg, (i.saturating_sub(8)) % per_msg);
            panic!("{name}: fs_proof differs first at byte {i}: round {} ({}), expected {:02x} got {:02x}", round + 1,
                   if off < 8 { "length prefix".to_string() } else { format!("evaluation {}", (off - 8) / 32) }, want[i], got[i]);
        }
        // the challenges (F::rand over the transcript) and the claimed sum
        let want_r = frs(&case["fs_randomness"]);
        assert_eq!(state.randomness.len(), want_r.len(), "{name}: randomness length");
        for (i, (a, b)) in state.randomness.iter().zip(&want_r).enumerate() {
            assert_eq!(fr_hex(a), fr_hex(b), "{name}: fs_randomness[{i}] (F::rand / transcript)");
        }
        assert_eq!(fr_hex(&MLSumcheck::extract_sum(&proof)), case["sum"].as_str().unwrap(), "{name}: extract_sum");
        let sub = MLSumcheck::verify(&poly.info(), fr(&case["sum"]), &proof).unwrap();
        assert_eq!(fr_hex(&sub.expected_evaluation), case["subclaim_expected"].as_str().unwrap(), "{name}: subclaim expected_evaluation");
        assert_eq!(fr_hex(&poly.evaluate(&sub.point)), case["subclaim_expected"].as_str().unwrap(), "{name}: evaluate(point)");
        println!("ml_{name}: proof bytes, randomness, sum and sub-claim match");
    }
}

/// the hot path with the transcript taken out: `prove_round` driven by the fixture's fixed challenges
#[test]
fn ml_interactive_vectors() {
    for name in ["nv1_trivial", "nv2_single", "nv3_c1shape", "nv5_deg12", "nv6_c3shape", "nv6_shared", "nv7_c2shape", "nv8_bench"] {
        let case = load(&format!("ml_{name}"));
        assert_eq!(case["name"].as_str().unwrap(), name, "{name}: fixture name");
        let nv = case["nv"].as_u64().unwrap() as usize;
        assert_eq!(case["tables"].as_array().unwrap().len() as u64, case["n_tables"].as_u64().unwrap(), "{name}: n_tables");
        // the in-memory form of a field element: Fp<MontBackend<FrConfig, 4>, 4>(BigInt([u64; 4])) holds the MONTGOMERY limbs, little-endian
        for (u, t) in case["tables"].as_array().unwrap().iter().enumerate() {
            let x = fr(&t[0]);
            let want: Vec<u64> = case["tables_mont0"][u].as_array().unwrap().iter().map(|l| l.as_u64().unwrap()).collect();
            assert_eq!((x.0).0.to_vec(), want, "{name}: Montgomery limbs of tables[{u}][0] (the C ABI's element layout)");
        }
        let poly = build_poly(&case);
        let mut state = IPForMLSumcheck::prover_init(&poly);
        let chal = frs(&case["challenges"]);
        let mut v_msg: Option<VerifierMsg<Fr>> = None;
        for i in 0..nv {
            let msg = IPForMLSumcheck::prove_round(&mut state, &v_msg);
            let want = frs(&case["rounds"][i]);
            assert_eq!(msg.evaluations.len(), want.len(), "{name}: round {} message length", i + 1);
            for (t, (a, b)) in msg.evaluations.iter().zip(&want).enumerate() {
                assert_eq!(fr_hex(a), fr_hex(b), "{name}: round {} evaluation {t} (prove_round)", i + 1);
            }
            v_msg = Some(VerifierMsg { randomness: chal[i] });
        }
        // what prove_round left behind: every flattened table bound nv - 1 times (two entries each), in first-occurrence order
        let ids: Vec<usize> = case["flattened_table_ids"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect();
        assert_eq!(state.flattened_ml_extensions.len(), ids.len(), "{name}: flattened tables");
        for (j, t) in state.flattened_ml_extensions.iter().enumerate() {
            let want = frs(&case["final_tables"][j]);
            assert_eq!(t.evaluations.len(), want.len(), "{name}: final table {j} (table {}) length", ids[j]);
            for (e, (a, b)) in t.evaluations.iter().zip(&want).enumerate() {
                assert_eq!(fr_hex(a), fr_hex(b), "{name}: final table {j} entry {e} (DenseMultilinearExtension::fix_variables)");
            }
        }
        println!("ml_{name}: interactive rounds and bound tables match");
    }
}

#[test]
fn gkr_vectors() {
    for dim in [2usize, 4, 6] {
        let g = load(&format!("gkr_dim{dim}"));
        let f1_pairs: Vec<(usize, Fr)> =
            g["f1_idx"].as_array().unwrap().iter().zip(g["f1_vals"].as_array().unwrap()).map(|(i, v)| (i.as_u64().unwrap() as usize, fr(v))).collect();
        let f1 = SparseMultilinearExtension::from_evaluations(3 * dim, &f1_pairs);
        let f2 = DenseMultilinearExtension::from_evaluations_vec(dim, frs(&g["f2"]));
        let f3 = DenseMultilinearExtension::from_evaluations_vec(dim, frs(&g["f3"]));
        let gg = frs(&g["g"]);
        // initialize_phase_one: the dense h_g and the sparse f1(g, ., .) with exactly the fixture's keys (zero values included)
        let (h_g, f1_g) = initialize_phase_one(&f1, &f3, &gg);
        for (i, (a, b)) in h_g.evaluations.iter().zip(frs(&g["h_g"])).enumerate() {
            assert_eq!(fr_hex(a), fr_hex(&b), "gkr dim {dim}: h_g[{i}]");
        }
        let got_keys: Vec<u64> = f1_g.evaluations.iter().map(|(k, _)| *k as u64).collect();
        let want_keys: Vec<u64> = g["f1_g_idx"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap()).collect();
        assert_eq!(got_keys, want_keys, "gkr dim {dim}: keys of f1_g (sparse fix_variables keeps / drops other entries than assumed)");
        for ((k, a), b) in f1_g.evaluations.iter().zip(frs(&g["f1_g_vals"])) {
            assert_eq!(fr_hex(a), fr_hex(&b), "gkr dim {dim}: f1_g[{k}]");
        }
        let u = frs(&g["u"]);
        let f1_gu = initialize_phase_two(&f1_g, &u);
        for (i, (a, b)) in f1_gu.evaluations.iter().zip(frs(&g["f1_gu"])).enumerate() {
            assert_eq!(fr_hex(a), fr_hex(&b), "gkr dim {dim}: f1_gu[{i}]");
        }
        // the whole proof: its messages are crate-private, but the verifier's sub-claim (u, v, expected) is a function of all of them
        let proof = GKRRoundSumcheck::prove(&mut Blake2b512Rng::setup(), &f1, &f2, &f3, &gg);
        assert_eq!(fr_hex(&proof.extract_sum()), g["sum"].as_str().unwrap(), "gkr dim {dim}: extract_sum");
        let sub = GKRRoundSumcheck::verify(&mut Blake2b512Rng::setup(), dim, &proof, fr(&g["sum"])).unwrap();
        for (i, (a, b)) in sub.u.iter().zip(&u).enumerate() {
            assert_eq!(fr_hex(a), fr_hex(b), "gkr dim {dim}: u[{i}] (phase-one messages / transcript)");
        }
        for (i, (a, b)) in sub.v.iter().zip(frs(&g["v"])).enumerate() {
            assert_eq!(fr_hex(a), fr_hex(&b), "gkr dim {dim}: v[{i}] (phase-two messages / transcript)");
        }
        assert_eq!(fr_hex(&sub.expected_evaluation), g["expected"].as_str().unwrap(), "gkr dim {dim}: expected_evaluation");
        assert!(sub.verify_subclaim(&f1, &f2, &f3, &gg), "gkr dim {dim}: verify_subclaim");
        // the messages of both phases: GKRRoundSumcheck::prove's loop (gkr_round_sumcheck/mod.rs:100-139) from its public parts
        assert_eq!(g["dim"].as_u64().unwrap() as usize, dim, "gkr dim {dim}: fixture dim");
        let mut rng = Blake2b512Rng::setup();
        let mut replay = |mut state: ark_sumcheck::ml_sumcheck::protocol::prover::ProverState<Fr>, want: &Value, key: &str| -> Vec<Fr> {
            let mut v_msg: Option<VerifierMsg<Fr>> = None;
            let mut point = Vec::new();
            for i in 0..dim {
                let msg = IPForMLSumcheck::prove_round(&mut state, &v_msg);
                for (t, (a, b)) in msg.evaluations.iter().zip(frs(&want[i])).enumerate() {
                    assert_eq!(fr_hex(a), fr_hex(&b), "gkr dim {dim}: {key} round {} evaluation {t}", i + 1);
                }
                rng.feed(&msg).unwrap();
                let r = IPForMLSumcheck::sample_round(&mut rng);
                point.push(r.randomness);
                v_msg = Some(r);
            }
            point
        };
        let u2 = replay(start_phase1_sumcheck(&h_g, &f2), &g["phase1"], "phase1");
        assert_eq!(u2.iter().map(fr_hex).collect::<Vec<_>>(), u.iter().map(fr_hex).collect::<Vec<_>>(), "gkr dim {dim}: replayed u");
        let _ = replay(start_phase2_sumcheck(&f1_gu, &f3, f2.evaluate(&u2)), &g["phase2"], "phase2");
        println!("gkr_dim{dim}: phase-one / phase-two initialisation, sum and sub-claim match");
    }
}
