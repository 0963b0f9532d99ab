//! Regenerates the golden fixtures of tests/golden/ from REAL arkworks (ark-sumcheck + ark-ff + ark-poly) so the
//! "parity unpinned" layer (F::rand, CanonicalSerialize, SparseMultilinearExtension::fix_variables) can be pinned on a
//! machine that has cargo.  Run: `cargo test --release -- --nocapture dump_vectors > vectors.json`, then diff against
//! tests/golden/ml_*.json ("fs_proof", "fs_randomness") with tools of your choice.  NOT BUILT in this repository's image.
use ark_ff::PrimeField;
use ark_poly::DenseMultilinearExtension;
use ark_std::rc::Rc;
use ark_sumcheck::ml_sumcheck::{data_structures::ListOfProductsOfPolynomials, MLSumcheck};
use ark_test_curves::bls12_381::Fr;

fn fr_from_hex(s: &str) -> Fr {
    let bytes: Vec<u8> = (0..s.len()).step_by(2).map(|i| u8::from_str_radix(&s[i..i + 2], 16).unwrap()).collect();
    Fr::from_be_bytes_mod_order(&bytes) // fixtures store canonical integers, big-endian hex
}

#[test]
fn dump_vectors() {
    for name in ["nv1_trivial", "nv2_single", "nv3_c1shape", "nv6_shared", "nv6_c3shape", "nv7_c2shape", "nv5_deg12", "nv8_bench"] {
        let text = std::fs::read_to_string(format!("../tests/golden/ml_{name}.json")).unwrap();
        let case: serde_json::Value = serde_json::from_str(&text).unwrap();
        let nv = case["nv"].as_u64().unwrap() as usize;
        let tables: Vec<Rc<DenseMultilinearExtension<Fr>>> = case["tables"]
            .as_array().unwrap().iter()
            .map(|t| Rc::new(DenseMultilinearExtension::from_evaluations_vec(nv, t.as_array().unwrap().iter().map(|x| fr_from_hex(x.as_str().unwrap())).collect())))
            .collect();
        let mut poly = ListOfProductsOfPolynomials::new(nv);
        for (k, shape) in case["shapes"].as_array().unwrap().iter().enumerate() {
            let c = fr_from_hex(case["products"][k][0].as_str().unwrap());
            poly.add_product(shape.as_array().unwrap().iter().map(|i| tables[i.as_u64().unwrap() as usize].clone()), c);
        }
        let proof = MLSumcheck::prove(&poly).unwrap();
        // compare against case["fs_proof"]: any mismatch pins down which recalled semantic is wrong
        let sum = MLSumcheck::extract_sum(&proof);
        assert_eq!(sum, fr_from_hex(case["sum"].as_str().unwrap()), "{name}: extract_sum");
        println!("{name}: extract_sum matches; serialise `proof` with ark-serialize and diff with fs_proof");
    }
}
