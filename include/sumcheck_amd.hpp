// sumcheck_amd.hpp -- header-only C++17 host-side mirror of arkworks-rs/sumcheck's public interface over the C ABI of
// libsumcheck_hip.so (include/sumcheck_hip.h).  Same names, argument meaning and error behaviour as the reference:
//
//   ListOfProductsOfPolynomials, PolynomialInfo      reference src/ml_sumcheck/data_structures.rs:25-110
//   ProverState, ProverMsg, IPForMLSumcheck          reference src/ml_sumcheck/protocol/prover.rs:13-153
//   VerifierMsg, SubClaim                            reference src/ml_sumcheck/protocol/verifier.rs:10-34
//   MLSumcheck                                       reference src/ml_sumcheck/mod.rs:18-101
//   Blake2b512Rng                                    reference src/rng.rs:22-81
//   SparseMultilinearExtension, GKRRoundSumcheck     reference src/gkr_round_sumcheck/mod.rs:22-193
//
// The reference is Rust; this is the compiled-language host layer for an image without a Rust toolchain (the Rust
// binding itself is rust-shim/).  The reference's panic!s surface as sumcheck::Panic carrying the same message;
// verifier rejection as sumcheck::Reject.  All prove_round work happens on the GPU inside the library.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "sumcheck_hip.h"

namespace sumcheck {

struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct Reject : std::runtime_error { // crate::Error::Reject
    explicit Reject(const std::string &m) : std::runtime_error(m) {}
};
inline void check(int rc) {
    if (rc == SC_OK) return;
    const std::string msg = sc_last_error();
    if (rc == SC_ERR_REJECT) throw Reject(msg);
    throw Panic(rc, msg);
}

// BLS12-381 Fr: 4 x u64 Montgomery limbs, the layout of ark_ff::Fp<MontBackend<FrConfig,4>,4>
struct Fr {
    uint64_t l[4] = {0, 0, 0, 0};
    bool operator==(const Fr &o) const { return std::memcmp(l, o.l, 32) == 0; }
    bool operator!=(const Fr &o) const { return !(*this == o); }
    static Fr one() { return Fr{{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}}; }
    static Fr zero() { return Fr{}; }
    // a valid ark_ff::Fp holds a residue < p; anything else is not a field element (operator+ below relies on it)
    bool is_canonical() const {
        static const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
        for (int i = 3; i >= 0; --i)
            if (l[i] != P[i]) return l[i] < P[i];
        return false;
    }
    // a + b mod p (host; only extract_sum needs it)
    friend Fr operator+(const Fr &a, const Fr &b) {
        static const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
        Fr r;
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (unsigned __int128)a.l[i] + b.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        bool ge = true;
        for (int i = 3; i >= 0; --i)
            if (r.l[i] != P[i]) {
                ge = r.l[i] > P[i];
                break;
            }
        if (ge) {
            uint64_t borrow = 0;
            for (int i = 0; i < 4; ++i) {
                unsigned __int128 d = (unsigned __int128)r.l[i] - P[i] - borrow;
                r.l[i] = (uint64_t)d;
                borrow = (uint64_t)(d >> 64) & 1;
            }
        }
        return r;
    }
};
static_assert(sizeof(Fr) == 32, "Fr must be 4 x u64");

class Blake2b512Rng { // FeedableRNG + RngCore
  public:
    Blake2b512Rng() : h_(sc_rng_setup()) {}
    static Blake2b512Rng setup() { return Blake2b512Rng(); }
    Blake2b512Rng(Blake2b512Rng &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Blake2b512Rng(const Blake2b512Rng &) = delete;
    ~Blake2b512Rng() {
        if (h_) sc_rng_free(h_);
    }
    void feed(const void *bytes, size_t len) { sc_rng_feed_bytes(h_, static_cast<const uint8_t *>(bytes), len); }
    void feed(const std::string &s) { feed(s.data(), s.size()); }
    void fill_bytes(uint8_t *dest, size_t len) { sc_rng_fill_bytes(h_, dest, len); }
    uint64_t next_u64() {
        uint8_t t[8];
        fill_bytes(t, 8);
        uint64_t x;
        std::memcpy(&x, t, 8);
        return x;
    }
    Fr rand_fr() { // F::rand(rng)
        Fr r;
        sc_rng_sample_fr(h_, r.l);
        return r;
    }
    sc_rng *raw() { return h_; }

  private:
    sc_rng *h_;
};

struct DenseMultilinearExtension {
    size_t num_vars = 0;
    std::vector<Fr> evaluations;
    static DenseMultilinearExtension from_evaluations_vec(size_t num_vars, std::vector<Fr> ev) {
        if (ev.size() != (size_t(1) << num_vars)) throw Panic(SC_ERR_BAD_ARG, "The size of evaluations should be 2^num_vars.");
        return DenseMultilinearExtension{num_vars, std::move(ev)};
    }
    static DenseMultilinearExtension rand(size_t num_vars, Blake2b512Rng &rng) {
        std::vector<Fr> ev(size_t(1) << num_vars);
        for (auto &x : ev) x = rng.rand_fr();
        return DenseMultilinearExtension{num_vars, std::move(ev)};
    }
    const Fr &operator[](size_t i) const { return evaluations[i]; }
    DenseMultilinearExtension fix_variables(const std::vector<Fr> &partial_point) const {
        if (partial_point.size() > num_vars) throw Panic(SC_ERR_BAD_ARG, "invalid partial point dimension");
        DenseMultilinearExtension out{num_vars - partial_point.size(), std::vector<Fr>(size_t(1) << (num_vars - partial_point.size()))};
        check(sc_fix_variables(evaluations[0].l, (uint32_t)num_vars, partial_point.empty() ? nullptr : partial_point[0].l,
                               (uint32_t)partial_point.size(), out.evaluations[0].l, 0));
        return out;
    }
    Fr evaluate(const std::vector<Fr> &point) const {
        if (point.size() != num_vars) throw Panic(SC_ERR_BAD_ARG, "invalid size of partial point");
        return fix_variables(point).evaluations[0];
    }
};

struct PolynomialInfo {
    size_t max_multiplicands = 0, num_variables = 0;
};

class ListOfProductsOfPolynomials {
  public:
    size_t max_multiplicands = 0;
    size_t num_variables;
    std::vector<std::pair<Fr, std::vector<size_t>>> products;
    std::vector<std::shared_ptr<DenseMultilinearExtension>> flattened_ml_extensions;

    explicit ListOfProductsOfPolynomials(size_t nv) : num_variables(nv) {}
    static ListOfProductsOfPolynomials new_(size_t nv) { return ListOfProductsOfPolynomials(nv); }
    PolynomialInfo info() const { return PolynomialInfo{max_multiplicands, num_variables}; }
    // data_structures.rs:71-96: multiplicands shared by pointer are stored once
    void add_product(const std::vector<std::shared_ptr<DenseMultilinearExtension>> &product, const Fr &coefficient) {
        if (product.empty()) throw Panic(SC_ERR_BAD_ARG, "assertion failed: !product.is_empty()");
        max_multiplicands = std::max(max_multiplicands, product.size());
        std::vector<size_t> indexed;
        for (const auto &m : product) {
            if (m->num_vars != num_variables) throw Panic(SC_ERR_BAD_ARG, "product has a multiplicand with wrong number of variables");
            auto it = lookup_.find(m.get());
            if (it != lookup_.end()) {
                indexed.push_back(it->second);
            } else {
                const size_t idx = flattened_ml_extensions.size();
                flattened_ml_extensions.push_back(m);
                lookup_[m.get()] = idx;
                indexed.push_back(idx);
            }
        }
        products.emplace_back(coefficient, std::move(indexed));
    }

    // marshalled view for the C ABI (valid while *this is alive and unchanged)
    struct Desc {
        sc_poly_desc d;
        std::vector<Fr> coeffs;
        std::vector<uint32_t> offsets, indices;
        std::vector<const uint64_t *> tables;
    };
    std::unique_ptr<Desc> desc() const {
        auto D = std::make_unique<Desc>();
        D->offsets.push_back(0);
        for (const auto &pr : products) {
            D->coeffs.push_back(pr.first);
            for (size_t i : pr.second) D->indices.push_back((uint32_t)i);
            D->offsets.push_back((uint32_t)D->indices.size());
        }
        for (const auto &t : flattened_ml_extensions) D->tables.push_back(t->evaluations[0].l);
        std::memset(&D->d, 0, sizeof(D->d));
        D->d.num_vars = (uint32_t)num_variables;
        D->d.max_multiplicands = (uint32_t)max_multiplicands;
        D->d.n_products = (uint32_t)products.size();
        D->d.coeffs = D->coeffs.empty() ? nullptr : D->coeffs[0].l;
        D->d.prod_offsets = D->offsets.data();
        D->d.prod_indices = D->indices.data();
        D->d.n_tables = (uint32_t)D->tables.size();
        D->d.tables = D->tables.data();
        return D;
    }

  private:
    std::unordered_map<const DenseMultilinearExtension *, size_t> lookup_;
};

struct ProverMsg {
    std::vector<Fr> evaluations;
};
struct VerifierMsg {
    Fr randomness;
};
struct SubClaim {
    std::vector<Fr> point;
    Fr expected_evaluation;
};
using Proof = std::vector<ProverMsg>;

class ProverState { // prover.rs:19-33, tables resident in HBM
  public:
    size_t num_vars = 0, max_multiplicands = 0;
    std::vector<std::pair<Fr, std::vector<size_t>>> list_of_products;
    ProverState() = default;
    ProverState(sc_prover *h, const ListOfProductsOfPolynomials &p)
        : num_vars(p.num_variables), max_multiplicands(p.max_multiplicands), list_of_products(p.products), h_(h),
          n_tables_(p.flattened_ml_extensions.size()) {}
    ProverState(ProverState &&o) noexcept { *this = std::move(o); }
    ProverState &operator=(ProverState &&o) noexcept {
        if (h_) sc_prover_free(h_);
        num_vars = o.num_vars;
        max_multiplicands = o.max_multiplicands;
        list_of_products = std::move(o.list_of_products);
        h_ = o.h_;
        n_tables_ = o.n_tables_;
        o.h_ = nullptr;
        return *this;
    }
    ~ProverState() {
        if (h_) sc_prover_free(h_);
    }
    size_t round() const {
        uint32_t r = 0;
        check(sc_prover_state(h_, nullptr, nullptr, nullptr, &r));
        return r;
    }
    std::vector<Fr> randomness() const {
        std::vector<Fr> buf(num_vars + 1);
        uint32_t n = 0;
        check(sc_prover_state(h_, buf[0].l, &n, nullptr, nullptr));
        buf.resize(n);
        return buf;
    }
    std::vector<DenseMultilinearExtension> flattened_ml_extensions() const {
        const size_t r = round();
        const size_t nv = num_vars - (r > 0 ? r - 1 : 0);
        std::vector<Fr> buf(n_tables_ << nv);
        check(sc_prover_state(h_, nullptr, nullptr, buf[0].l, nullptr));
        std::vector<DenseMultilinearExtension> out;
        for (size_t u = 0; u < n_tables_; ++u)
            out.push_back(DenseMultilinearExtension{nv, std::vector<Fr>(buf.begin() + (u << nv), buf.begin() + ((u + 1) << nv))});
        return out;
    }
    sc_prover *raw() { return h_; }
    // library policy (no reference counterpart): false = no kernel of this handle ever waits for the host (a host with HIP streams of its
    // own: see the interference contract in sumcheck_hip.h); patience of the resident kernel behind prove_round called round by round
    void set_polling(bool allow) { check(sc_prover_set_polling(h_, allow ? 1 : 0)); }
    void set_resident(uint32_t patience_polls) { check(sc_prover_set_resident(h_, patience_polls)); }

  private:
    sc_prover *h_ = nullptr;
    size_t n_tables_ = 0;
};

// The library keeps device memory between calls (the last prover it built, the work areas of evaluate / fix_variables, the GKR
// scratch) so that one-shot calls cost what kept state costs; this returns all of it.
inline void release_caches() { check(sc_release_caches()); }
inline void set_cache_limit(uint64_t bytes) { check(sc_set_cache_limit(bytes)); } // what each of those caches may keep; 0 = nothing

struct IPForMLSumcheck {
    static ProverState prover_init(const ListOfProductsOfPolynomials &polynomial) { // prover.rs:49-69
        auto D = polynomial.desc();
        sc_prover *h = nullptr;
        check(sc_prover_init(&D->d, &h));
        return ProverState(h, polynomial);
    }
    static ProverMsg prove_round(ProverState &state, const std::optional<VerifierMsg> &v_msg) { // prover.rs:74-153
        ProverMsg m;
        m.evaluations.resize(state.max_multiplicands + 1);
        check(sc_prove_round(state.raw(), v_msg ? v_msg->randomness.l : nullptr, m.evaluations[0].l));
        return m;
    }
    static VerifierMsg sample_round(Blake2b512Rng &rng) { return VerifierMsg{rng.rand_fr()}; } // verifier.rs:128-131
};

struct MLSumcheck {
    static Fr extract_sum(const Proof &proof) { return proof[0].evaluations[0] + proof[0].evaluations[1]; } // mod.rs:26-28
    static std::pair<Proof, ProverState> prove_as_subprotocol(Blake2b512Rng &fs_rng, const ListOfProductsOfPolynomials &polynomial) {
        auto D = polynomial.desc();
        const size_t nv = polynomial.num_variables, Dg = polynomial.max_multiplicands + 1;
        std::vector<Fr> flat(std::max<size_t>(nv, 1) * Dg);
        sc_prover *h = nullptr;
        check(sc_ml_prove(&D->d, fs_rng.raw(), flat[0].l, &h));
        Proof proof(nv);
        for (size_t i = 0; i < nv; ++i) proof[i].evaluations.assign(flat.begin() + i * Dg, flat.begin() + (i + 1) * Dg);
        return {std::move(proof), ProverState(h, polynomial)};
    }
    static Proof prove(const ListOfProductsOfPolynomials &polynomial) { // mod.rs:42-45
        // no state comes back: the library reads device tables in place and keeps the prover for the next proof of this shape
        auto D = polynomial.desc();
        const size_t nv = polynomial.num_variables, Dg = polynomial.max_multiplicands + 1;
        std::vector<Fr> flat(std::max<size_t>(nv, 1) * Dg);
        check(sc_ml_prove(&D->d, nullptr, flat[0].l, nullptr));
        Proof proof(nv);
        for (size_t i = 0; i < nv; ++i) proof[i].evaluations.assign(flat.begin() + i * Dg, flat.begin() + (i + 1) * Dg);
        return proof;
    }
    static SubClaim verify_as_subprotocol(Blake2b512Rng &fs_rng, const PolynomialInfo &info, const Fr &claimed_sum, const Proof &proof) {
        const size_t nv = info.num_variables, Dg = info.max_multiplicands + 1;
        if (proof.size() < nv) throw Panic(SC_ERR_BAD_ARG, "proof is incomplete");
        std::vector<Fr> flat(std::max<size_t>(nv, 1) * Dg);
        for (size_t i = 0; i < nv; ++i) {
            if (proof[i].evaluations.size() != Dg) throw Panic(SC_ERR_BAD_ARG, "incorrect number of evaluations");
            std::copy(proof[i].evaluations.begin(), proof[i].evaluations.end(), flat.begin() + i * Dg);
        }
        SubClaim sub;
        sub.point.resize(std::max<size_t>(nv, 1));
        check(sc_ml_verify((uint32_t)nv, (uint32_t)info.max_multiplicands, claimed_sum.l, flat[0].l, (uint64_t)(nv * Dg), fs_rng.raw(), sub.point[0].l,
                           sub.expected_evaluation.l));
        sub.point.resize(nv);
        return sub;
    }
    static SubClaim verify(const PolynomialInfo &info, const Fr &claimed_sum, const Proof &proof) { // mod.rs:73-80
        Blake2b512Rng rng;
        return verify_as_subprotocol(rng, info, claimed_sum, proof);
    }
};

// ListOfProductsOfPolynomials::evaluate (data_structures.rs:99-109): one library call, all tables folded on the GPU
inline Fr evaluate(const ListOfProductsOfPolynomials &poly, const std::vector<Fr> &point) {
    if (point.size() != poly.num_variables) throw Panic(SC_ERR_BAD_ARG, "wrong number of variables");
    auto D = poly.desc();
    Fr out;
    check(sc_poly_evaluate(&D->d, point.empty() ? nullptr : point[0].l, out.l, nullptr));
    return out;
}

struct SparseMultilinearExtension {
    size_t num_vars = 0;
    std::vector<uint64_t> indices; // distinct
    std::vector<Fr> values;
};

struct GKRProof {
    Proof phase1_sumcheck_msgs, phase2_sumcheck_msgs;
    Fr extract_sum() const { return phase1_sumcheck_msgs[0].evaluations[0] + phase1_sumcheck_msgs[0].evaluations[1]; }
};

// initialize_phase_one (gkr_round_sumcheck/mod.rs:22-42)
inline std::pair<DenseMultilinearExtension, SparseMultilinearExtension> initialize_phase_one(const SparseMultilinearExtension &f1,
                                                                                             const DenseMultilinearExtension &f3,
                                                                                             const std::vector<Fr> &g) {
    const size_t dim = f3.num_vars;
    if (f1.num_vars != 3 * dim || g.size() != dim) throw Panic(SC_ERR_BAD_ARG, "assertion failed: dimensions");
    DenseMultilinearExtension hg{dim, std::vector<Fr>(size_t(1) << dim)};
    SparseMultilinearExtension f1g{2 * dim, std::vector<uint64_t>(std::max<size_t>(f1.indices.size(), 1)), std::vector<Fr>(std::max<size_t>(f1.indices.size(), 1))};
    uint64_t n1 = 0;
    check(sc_gkr_phase_one(f1.indices.data(), f1.values.empty() ? nullptr : f1.values[0].l, f1.indices.size(), (uint32_t)dim, f3.evaluations[0].l,
                           g[0].l, 0, hg.evaluations[0].l, f1g.indices.data(), f1g.values[0].l, &n1));
    f1g.indices.resize(n1);
    f1g.values.resize(n1);
    return {std::move(hg), std::move(f1g)};
}
// initialize_phase_two (gkr_round_sumcheck/mod.rs:57-63)
inline DenseMultilinearExtension initialize_phase_two(const SparseMultilinearExtension &f1_g, const std::vector<Fr> &u) {
    if (u.size() * 2 != f1_g.num_vars) throw Panic(SC_ERR_BAD_ARG, "assertion failed: u.len() * 2 == f1_g.num_vars");
    DenseMultilinearExtension out{u.size(), std::vector<Fr>(size_t(1) << u.size())};
    check(sc_gkr_phase_two(f1_g.indices.data(), f1_g.values.empty() ? nullptr : f1_g.values[0].l, f1_g.indices.size(), (uint32_t)u.size(), u[0].l,
                           0, out.evaluations[0].l));
    return out;
}

// gkr_round_sumcheck/data_structures.rs:22-56
struct GKRRoundSumcheckSubClaim {
    std::vector<Fr> u, v;
    Fr expected_evaluation;
    // expected_evaluation == f1(g, u, v) * f2(u) * f3(v); the three oracle queries run on the GPU
    bool verify_subclaim(const SparseMultilinearExtension &f1, const DenseMultilinearExtension &f2, const DenseMultilinearExtension &f3,
                         const std::vector<Fr> &g) const {
        const size_t dim = u.size();
        if (v.size() != dim || g.size() != dim || f1.num_vars != 3 * dim || f2.num_vars != dim || f3.num_vars != dim)
            throw Panic(SC_ERR_BAD_ARG, "assertion failed: dimensions");
        std::vector<Fr> guv(g);
        guv.insert(guv.end(), u.begin(), u.end());
        guv.insert(guv.end(), v.begin(), v.end());
        Fr a, ab, abc;
        check(sc_sparse_evaluate(f1.indices.data(), f1.values.empty() ? nullptr : f1.values[0].l, f1.indices.size(), (uint32_t)f1.num_vars,
                                 guv.empty() ? nullptr : guv[0].l, a.l));
        const Fr b = f2.evaluate(u), c = f3.evaluate(v);
        check(sc_fr_elementwise(0, a.l, b.l, ab.l, 1));
        check(sc_fr_elementwise(0, ab.l, c.l, abc.l, 1));
        return abc == expected_evaluation;
    }
};

struct GKRRoundSumcheck {
    // verifier_init{max_multiplicands: 2} + (feed, verify_round) x dim + check_and_generate_subclaim (mod.rs:157-166, 173-182)
    static std::pair<std::vector<Fr>, Fr> verify_phase(Blake2b512Rng &rng, const Proof &msgs, size_t dim, const Fr &asserted_sum) {
        if (msgs.size() < dim) throw Panic(SC_ERR_BAD_ARG, "proof is incomplete");
        std::vector<Fr> rs;
        for (size_t i = 0; i < dim; ++i) {
            if (msgs[i].evaluations.empty()) throw Panic(SC_ERR_BAD_ARG, "incorrect number of evaluations");
            sc_rng_feed_prover_msg(rng.raw(), msgs[i].evaluations[0].l, (uint32_t)msgs[i].evaluations.size());
            rs.push_back(rng.rand_fr());
        }
        Fr expected = asserted_sum;
        if (!expected.is_canonical()) throw Panic(SC_ERR_BAD_ARG, "claimed sum is not a canonical field element");
        for (size_t i = 0; i < dim; ++i) {
            const auto &ev = msgs[i].evaluations;
            if (ev.size() != 3) throw Panic(SC_ERR_BAD_ARG, "incorrect number of evaluations");
            for (const Fr &e : ev) // the raw-limb sum below is only a field addition on canonical operands
                if (!e.is_canonical()) throw Panic(SC_ERR_BAD_ARG, "proof element is not a canonical field element");
            if (ev[0] + ev[1] != expected) throw Reject("Prover message is not consistent with the claim.");
            check(sc_interpolate_uni_poly(ev[0].l, 3, rs[i].l, expected.l));
        }
        return {rs, expected};
    }
    static GKRRoundSumcheckSubClaim verify(Blake2b512Rng &rng, size_t f2_num_vars, const GKRProof &proof, const Fr &claimed_sum) { // mod.rs:147-192
        auto p1 = verify_phase(rng, proof.phase1_sumcheck_msgs, f2_num_vars, claimed_sum);
        auto p2 = verify_phase(rng, proof.phase2_sumcheck_msgs, f2_num_vars, p1.second);
        return GKRRoundSumcheckSubClaim{std::move(p1.first), std::move(p2.first), p2.second};
    }
    static GKRProof prove(Blake2b512Rng &rng, const SparseMultilinearExtension &f1, const DenseMultilinearExtension &f2,
                          const DenseMultilinearExtension &f3, const std::vector<Fr> &g) { // gkr_round_sumcheck/mod.rs:93-139
        const size_t dim = f2.num_vars;
        if (f1.num_vars != 3 * dim || f3.num_vars != dim || g.size() != dim) throw Panic(SC_ERR_BAD_ARG, "assertion failed: dimensions");
        std::vector<Fr> flat(2 * std::max<size_t>(dim, 1) * 3);
        check(sc_gkr_prove(rng.raw(), f1.indices.data(), f1.values.empty() ? nullptr : f1.values[0].l, f1.indices.size(), (uint32_t)dim,
                           f2.evaluations[0].l, f3.evaluations[0].l, g[0].l, 0, flat[0].l, nullptr));
        GKRProof pr;
        for (size_t i = 0; i < dim; ++i) {
            pr.phase1_sumcheck_msgs.push_back(ProverMsg{std::vector<Fr>(flat.begin() + 3 * i, flat.begin() + 3 * i + 3)});
            pr.phase2_sumcheck_msgs.push_back(ProverMsg{std::vector<Fr>(flat.begin() + 3 * (dim + i), flat.begin() + 3 * (dim + i) + 3)});
        }
        return pr;
    }
};

} // namespace sumcheck
