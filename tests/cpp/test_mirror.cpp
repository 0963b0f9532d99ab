// C++ mirror of the reference's own tests (src/ml_sumcheck/test.rs, src/gkr_round_sumcheck/test.rs) written against
// include/sumcheck_amd.hpp, i.e. through the C ABI of libsumcheck_hip.so.  Needs a GPU (run by tests/test_cpp_mirror.py).
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>

#include "sumcheck_amd.hpp"

using namespace sumcheck;

static int g_failed = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) {                                                        \
            std::printf("  FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
            ++g_failed;                                                       \
        }                                                                     \
    } while (0)

static std::vector<Fr> hadamard(const std::vector<Fr> &a, const std::vector<Fr> &b) {
    std::vector<Fr> out(a.size());
    check(sc_fr_elementwise(0, a[0].l, b[0].l, out[0].l, a.size()));
    return out;
}
static Fr mul(const Fr &a, const Fr &b) {
    Fr o;
    check(sc_fr_elementwise(0, a.l, b.l, o.l, 1));
    return o;
}
static Fr sum(const std::vector<Fr> &v) {
    Fr s = Fr::zero();
    for (const auto &x : v) s = s + x;
    return s;
}

// random_product / random_list_of_products (test.rs:15-62): the sum is computed independently of the prover
static std::pair<ListOfProductsOfPolynomials, Fr> random_list_of_products(size_t nv, size_t lo_m, size_t hi_m, size_t num_products,
                                                                           Blake2b512Rng &rng) {
    ListOfProductsOfPolynomials poly(nv);
    Fr total = Fr::zero();
    for (size_t k = 0; k < num_products; ++k) {
        const size_t m = lo_m + (size_t)(rng.next_u64() % (hi_m - lo_m));
        std::vector<std::shared_ptr<DenseMultilinearExtension>> mult;
        std::vector<Fr> prod;
        for (size_t j = 0; j < m; ++j) {
            mult.push_back(std::make_shared<DenseMultilinearExtension>(DenseMultilinearExtension::rand(nv, rng)));
            prod = j == 0 ? mult.back()->evaluations : hadamard(prod, mult.back()->evaluations);
        }
        const Fr c = rng.rand_fr();
        poly.add_product(mult, c);
        total = total + mul(sum(prod), c);
    }
    return {std::move(poly), total};
}

static void test_polynomial(size_t nv, size_t lo_m, size_t hi_m, size_t np, Blake2b512Rng &rng) { // test.rs:64-75
    auto [poly, asserted_sum] = random_list_of_products(nv, lo_m, hi_m, np, rng);
    const Proof proof = MLSumcheck::prove(poly);
    const SubClaim sub = MLSumcheck::verify(poly.info(), asserted_sum, proof);
    EXPECT(evaluate(poly, sub.point) == sub.expected_evaluation);
}
static void test_protocol(size_t nv, size_t lo_m, size_t hi_m, size_t np, Blake2b512Rng &rng) { // test.rs:77-97: hand-driven
    auto [poly, asserted_sum] = random_list_of_products(nv, lo_m, hi_m, np, rng);
    ProverState ps = IPForMLSumcheck::prover_init(poly);
    std::optional<VerifierMsg> vm;
    Proof msgs;
    std::vector<Fr> rs;
    for (size_t i = 0; i < poly.num_variables; ++i) {
        msgs.push_back(IPForMLSumcheck::prove_round(ps, vm));
        vm = IPForMLSumcheck::sample_round(rng); // a non Fiat-Shamir verifier
        rs.push_back(vm->randomness);
    }
    // check_and_generate_subclaim with the verifier's own randomness: P_i(0)+P_i(1) chain + interpolation
    Fr expected = asserted_sum;
    for (size_t i = 0; i < poly.num_variables; ++i) {
        EXPECT(msgs[i].evaluations[0] + msgs[i].evaluations[1] == expected);
        Fr nxt;
        check(sc_interpolate_uni_poly(msgs[i].evaluations[0].l, (uint32_t)msgs[i].evaluations.size(), rs[i].l, nxt.l));
        expected = nxt;
    }
    EXPECT(evaluate(poly, rs) == expected);
}
static void test_polynomial_as_subprotocol(size_t nv, size_t lo_m, size_t hi_m, size_t np, Blake2b512Rng &rng, const std::string &pl,
                                           const std::string &vl, bool expect_ok) { // test.rs:99-120
    auto [poly, asserted_sum] = random_list_of_products(nv, lo_m, hi_m, np, rng);
    Blake2b512Rng prng, vrng;
    prng.feed(pl);
    vrng.feed(vl);
    auto [proof, state] = MLSumcheck::prove_as_subprotocol(prng, poly);
    bool ok = true;
    try {
        const SubClaim sub = MLSumcheck::verify_as_subprotocol(vrng, poly.info(), asserted_sum, proof);
        ok = evaluate(poly, sub.point) == sub.expected_evaluation && state.randomness() == sub.point;
    } catch (const Reject &) {
        ok = false;
    }
    EXPECT(ok == expect_ok);
}

static void test_trivial_polynomial(Blake2b512Rng &rng) { // test.rs:122-144 (nv = 1, 4..12 multiplicands)
    for (int it = 0; it < 3; ++it) {
        test_polynomial(1, 4, 13, 5, rng);
        test_protocol(1, 4, 13, 5, rng);
        test_polynomial_as_subprotocol(1, 4, 13, 5, rng, "Test Trivial Works", "Test Trivial Works", true);
    }
}
static void test_normal_polynomial(Blake2b512Rng &rng) { // test.rs:145-167 (nv = 12, 4..8 multiplicands)
    for (int it = 0; it < 2; ++it) {
        test_polynomial(12, 4, 9, 5, rng);
        test_protocol(12, 4, 9, 5, rng);
        test_polynomial_as_subprotocol(12, 4, 9, 5, rng, "Test Trivial Works", "Test Trivial Works", true);
    }
    test_polynomial_as_subprotocol(12, 4, 9, 5, rng, "Test Trivial Works", "Test Trivial Fails", false); // test.rs:168-186
}
static void zero_polynomial_should_error(Blake2b512Rng &rng) { // test.rs:187-204
    bool panicked = false;
    try {
        test_polynomial(0, 4, 13, 5, rng);
    } catch (const Panic &e) {
        panicked = std::string(e.what()).find("Attempt to prove a constant.") != std::string::npos;
    }
    EXPECT(panicked);
    panicked = false;
    try {
        test_protocol(0, 4, 13, 5, rng);
    } catch (const Panic &e) {
        panicked = std::string(e.what()).find("Attempt to prove a constant.") != std::string::npos;
    }
    EXPECT(panicked);
}
static void test_extract_sum(Blake2b512Rng &rng) { // test.rs:206-213
    auto [poly, asserted_sum] = random_list_of_products(8, 3, 4, 3, rng);
    const Proof proof = MLSumcheck::prove(poly);
    EXPECT(MLSumcheck::extract_sum(proof) == asserted_sum);
}
static void test_shared_reference(Blake2b512Rng &rng) { // test.rs:215-269
    std::vector<std::shared_ptr<DenseMultilinearExtension>> ml;
    for (int i = 0; i < 5; ++i) ml.push_back(std::make_shared<DenseMultilinearExtension>(DenseMultilinearExtension::rand(8, rng)));
    ListOfProductsOfPolynomials poly(8);
    poly.add_product({ml[2], ml[3], ml[0]}, rng.rand_fr());
    poly.add_product({ml[1], ml[4], ml[4]}, rng.rand_fr());
    poly.add_product({ml[3], ml[2], ml[1]}, rng.rand_fr());
    poly.add_product({ml[0], ml[0]}, rng.rand_fr());
    poly.add_product({ml[4]}, rng.rand_fr());
    EXPECT(poly.flattened_ml_extensions.size() == 5);
    {
        ProverState prover = IPForMLSumcheck::prover_init(poly);
        EXPECT(prover.flattened_ml_extensions().size() == 5);
    }
    const Proof proof = MLSumcheck::prove(poly);
    const Fr asserted_sum = MLSumcheck::extract_sum(proof);
    const SubClaim sub = MLSumcheck::verify(poly.info(), asserted_sum, proof);
    EXPECT(evaluate(poly, sub.point) == sub.expected_evaluation);
}
static void test_prover_state_machine(Blake2b512Rng &rng) { // the panics of prover.rs:79-98
    auto pr = random_list_of_products(3, 2, 3, 1, rng);
    ProverState ps = IPForMLSumcheck::prover_init(pr.first);
    auto expect_panic = [&](std::function<void()> f, const char *msg) {
        bool ok = false;
        try {
            f();
        } catch (const Panic &e) {
            ok = std::string(e.what()).find(msg) != std::string::npos;
        }
        EXPECT(ok);
    };
    const VerifierMsg vm = IPForMLSumcheck::sample_round(rng);
    expect_panic([&] { IPForMLSumcheck::prove_round(ps, vm); }, "first round should be prover first.");
    IPForMLSumcheck::prove_round(ps, std::nullopt);
    expect_panic([&] { IPForMLSumcheck::prove_round(ps, std::nullopt); }, "verifier message is empty");
    IPForMLSumcheck::prove_round(ps, vm);
    IPForMLSumcheck::prove_round(ps, vm);
    expect_panic([&] { IPForMLSumcheck::prove_round(ps, vm); }, "Prover is not active");
}
static void test_gkr_extract(Blake2b512Rng &rng, size_t dim) { // gkr_round_sumcheck/test.rs:76-88
    SparseMultilinearExtension f1;
    f1.num_vars = 3 * dim;
    const uint64_t mask = (uint64_t(1) << (3 * dim)) - 1;
    std::unordered_map<uint64_t, int> seen;
    while (f1.indices.size() < (size_t(1) << dim)) { // rand_with_config(3*dim, 1 << dim)
        const uint64_t idx = rng.next_u64() & mask;
        if (seen.emplace(idx, 1).second) {
            f1.indices.push_back(idx);
            f1.values.push_back(rng.rand_fr());
        }
    }
    const DenseMultilinearExtension f2 = DenseMultilinearExtension::rand(dim, rng), f3 = DenseMultilinearExtension::rand(dim, rng);
    std::vector<Fr> g(dim);
    for (auto &x : g) x = rng.rand_fr();
    // naive sum: sum over the entries of f1(g, x, y) of v * f2[x] * f3[y]
    auto [hg, f1g] = initialize_phase_one(f1, f3, g);
    std::vector<Fr> a(f1g.indices.size()), b(f1g.indices.size());
    for (size_t i = 0; i < f1g.indices.size(); ++i) {
        a[i] = f2[f1g.indices[i] & ((uint64_t(1) << dim) - 1)];
        b[i] = f3[f1g.indices[i] >> dim];
    }
    const Fr expected = sum(hadamard(hadamard(f1g.values, a), b));
    EXPECT(sum(hadamard(hg.evaluations, f2.evaluations)) == expected); // h_g is consistent with its definition
    Blake2b512Rng fs;
    const GKRProof proof = GKRRoundSumcheck::prove(fs, f1, f2, f3, g);
    EXPECT(proof.extract_sum() == expected);
    EXPECT(proof.phase1_sumcheck_msgs.size() == dim && proof.phase2_sumcheck_msgs.size() == dim);
    // test_circuit (test.rs:57-69): verify with a fresh transcript, then the three oracle queries of the sub-claim
    Blake2b512Rng vs;
    const GKRRoundSumcheckSubClaim sub = GKRRoundSumcheck::verify(vs, dim, proof, expected);
    EXPECT(sub.u.size() == dim && sub.v.size() == dim);
    EXPECT(sub.verify_subclaim(f1, f2, f3, g));
    // a wrong claimed sum is rejected in the first round; a tampered sub-claim fails the oracle check
    bool rejected = false;
    try {
        Blake2b512Rng vs2;
        GKRRoundSumcheck::verify(vs2, dim, proof, expected + Fr::one());
    } catch (const Reject &) {
        rejected = true;
    }
    EXPECT(rejected);
    GKRRoundSumcheckSubClaim bad = sub;
    bad.expected_evaluation = bad.expected_evaluation + Fr::one();
    EXPECT(!bad.verify_subclaim(f1, f2, f3, g));
}

// A proof element or claimed sum >= p is not a field element (the reference's Fp cannot hold one): ev0 + p, ev1 + p hash to the honest
// transcript and, added as raw limbs, wrap past 2^256 -- every verifier entry must refuse the encoding before any arithmetic.
static void test_verifier_refuses_non_canonical(Blake2b512Rng &rng) {
    static const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    auto plus_p = [](Fr x) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (unsigned __int128)x.l[i] + P[i];
            x.l[i] = (uint64_t)c;
            c >>= 64;
        }
        return x;
    };
    auto refuses = [&](std::function<void()> f) {
        try {
            f();
        } catch (const Panic &e) {
            return std::string(e.what()).find("canonical") != std::string::npos;
        } catch (const Reject &) {
            return false; // must be refused as malformed, not merely rejected
        }
        return false;
    };
    auto [poly, asserted_sum] = random_list_of_products(5, 2, 4, 2, rng);
    const Proof proof = MLSumcheck::prove(poly);
    EXPECT(MLSumcheck::extract_sum(proof) == asserted_sum);
    Proof bad = proof;
    bad[0].evaluations[0] = plus_p(bad[0].evaluations[0]);
    bad[0].evaluations[1] = plus_p(bad[0].evaluations[1]);
    EXPECT(!bad[0].evaluations[0].is_canonical());
    EXPECT(refuses([&] { MLSumcheck::verify(poly.info(), asserted_sum, bad); }));
    EXPECT(refuses([&] { MLSumcheck::verify(poly.info(), plus_p(asserted_sum), proof); }));
    Proof shortp = proof; // a message one evaluation short: "incorrect number of evaluations" (verifier.rs:60-62)
    shortp[1].evaluations.pop_back();
    bool panicked = false;
    try {
        MLSumcheck::verify(poly.info(), asserted_sum, shortp);
    } catch (const Panic &e) {
        panicked = std::string(e.what()).find("incorrect number of evaluations") != std::string::npos;
    }
    EXPECT(panicked);
    // GKR verify_phase adds raw limbs in the mirror itself
    GKRProof gp;
    gp.phase1_sumcheck_msgs = {ProverMsg{{plus_p(Fr::one()), Fr::zero(), Fr::zero()}}};
    gp.phase2_sumcheck_msgs = {ProverMsg{{Fr::zero(), Fr::zero(), Fr::zero()}}};
    EXPECT(refuses([&] {
        Blake2b512Rng vs;
        GKRRoundSumcheck::verify(vs, 1, gp, Fr::one());
    }));
}

// library policy switches (no reference counterpart): the hand-driven protocol of test.rs:77-97 gives the same messages with the
// device-side waits off, with the resident kernel off, with nothing cached -- and through prove_as_subprotocol with polling off
static void test_policy_switches_do_not_change_a_proof(Blake2b512Rng &rng) {
    auto [poly, asserted_sum] = random_list_of_products(11, 2, 4, 3, rng);
    std::vector<Fr> chal;
    for (size_t i = 0; i < poly.num_variables; ++i) chal.push_back(rng.rand_fr());
    auto dialogue = [&](int mode) {
        ProverState ps = IPForMLSumcheck::prover_init(poly);
        if (mode == 1) ps.set_polling(false);
        if (mode == 2) ps.set_resident(0);
        Proof msgs;
        std::optional<VerifierMsg> vm;
        for (size_t i = 0; i < poly.num_variables; ++i) {
            msgs.push_back(IPForMLSumcheck::prove_round(ps, vm));
            vm = VerifierMsg{chal[i]};
            if (mode == 3 && i == 7) (void)ps.flattened_ml_extensions(); // asks the resident kernel to leave in the middle
        }
        return msgs;
    };
    const Proof ref = dialogue(0);
    EXPECT(ref[0].evaluations[0] + ref[0].evaluations[1] == asserted_sum);
    for (int mode = 1; mode <= 3; ++mode) {
        const Proof got = dialogue(mode);
        bool same = got.size() == ref.size();
        for (size_t i = 0; same && i < ref.size(); ++i) same = got[i].evaluations == ref[i].evaluations;
        EXPECT(same);
    }
    set_cache_limit(0); // nothing kept between calls
    const Proof a = MLSumcheck::prove(poly);
    set_cache_limit(16ull << 30);
    const Proof b = MLSumcheck::prove(poly);
    bool same = a.size() == b.size();
    for (size_t i = 0; same && i < a.size(); ++i) same = a[i].evaluations == b[i].evaluations;
    EXPECT(same);
    EXPECT(evaluate(poly, MLSumcheck::verify(poly.info(), asserted_sum, a).point) == MLSumcheck::verify(poly.info(), asserted_sum, a).expected_evaluation);
}

int main() {
    if (sc_device_count() <= 0) {
        std::printf("no HIP device: these tests need a GPU\n");
        return 2;
    }
    Blake2b512Rng rng; // deterministic test inputs: squeeze a labelled transcript
    rng.feed(std::string("sumcheck_amd C++ mirror tests"));
    struct T {
        const char *name;
        std::function<void()> fn;
    } tests[] = {
        {"test_trivial_polynomial", [&] { test_trivial_polynomial(rng); }},
        {"test_normal_polynomial (+ different transcripts fail)", [&] { test_normal_polynomial(rng); }},
        {"zero_polynomial_should_error", [&] { zero_polynomial_should_error(rng); }},
        {"test_extract_sum", [&] { test_extract_sum(rng); }},
        {"test_shared_reference", [&] { test_shared_reference(rng); }},
        {"prover state machine panics", [&] { test_prover_state_machine(rng); }},
        {"verifier refuses non-canonical encodings and short messages", [&] { test_verifier_refuses_non_canonical(rng); }},
        {"policy switches (polling, resident kernel, cache limit) do not change a proof", [&] { test_policy_switches_do_not_change_a_proof(rng); }},
        {"gkr test_extract (dim 6)", [&] { test_gkr_extract(rng, 6); }},
        {"gkr test_small shape (dim 9)", [&] { test_gkr_extract(rng, 9); }},
    };
    for (auto &t : tests) {
        const int before = g_failed;
        try {
            t.fn();
        } catch (const std::exception &e) {
            std::printf("  EXCEPTION in %s: %s\n", t.name, e.what());
            ++g_failed;
        }
        std::printf("%s %s\n", g_failed == before ? "ok    " : "FAILED", t.name);
    }
    std::printf("%s\n", g_failed ? "SOME TESTS FAILED" : "ALL TESTS PASSED");
    return g_failed ? 1 : 0;
}
