/* abi_smoke.c -- include/sumcheck_hip.h is a C header: this file is compiled as C99 with -pedantic -Werror, linked against
 * libsumcheck_hip.so and run by tests/test_host.py (no GPU needed: only host-side entry points are called, plus the loud failure of a
 * compute entry point when no device is visible).  What a cgo / JNI / ctypes binding would do first. */
#include <stdio.h>
#include <string.h>

#include "../../include/sumcheck_hip.h"

int main(void) {
    unsigned char out[64];
    uint64_t fr[4], claimed[4] = {0, 0, 0, 0}, proof[2 * 3 * 4], point[2 * 4], expected[4];
    sc_rng *rng;
    int rc;
    if (sc_abi_version() != SC_ABI_VERSION) {
        printf("ABI version mismatch: header %d, library %d\n", SC_ABI_VERSION, sc_abi_version());
        return 1;
    }
    rng = sc_rng_setup();
    if (!rng) return 2;
    sc_rng_feed_bytes(rng, (const uint8_t *)"abc", 3);
    sc_rng_fill_bytes(rng, out, 64); /* the first squeeze is BLAKE2b-512("abc") (RFC 7693 appendix A) */
    if (out[0] != 0xba || out[1] != 0x80 || out[2] != 0xa5 || out[63] != 0x23) {
        printf("transcript: unexpected digest %02x%02x..%02x\n", out[0], out[1], out[63]);
        return 3;
    }
    sc_rng_feed_poly_info(rng, 3, 2);
    sc_rng_sample_fr(rng, fr);
    if ((fr[3] >> 63) != 0) return 4; /* F::rand clears the top bit and rejects >= p */
    sc_rng_free(rng);
    /* an all-zero proof of the zero polynomial is accepted: P(0) + P(1) = 0 = claim in every round */
    memset(proof, 0, sizeof proof);
    rc = sc_ml_verify(2, 2, claimed, proof, 2 * 3, NULL, point, expected);
    if (rc != SC_OK) {
        printf("sc_ml_verify: %d %s\n", rc, sc_last_error());
        return 5;
    }
    /* a wrong element count is refused before anything is read */
    if (sc_ml_verify(2, 2, claimed, proof, 5, NULL, point, expected) != SC_ERR_BAD_ARG) return 6;
    if (sc_ml_verify(2, 0, claimed, proof, 2, NULL, point, expected) != SC_ERR_BAD_ARG) return 7;
    if (sc_device_count() == 0) { /* no CPU fallback: compute entry points fail loudly */
        sc_poly_desc d;
        sc_prover *p = NULL;
        const uint64_t *tabs[1];
        uint32_t offs[2] = {0, 1}, idx[1] = {0};
        uint64_t coeff[4] = {1, 0, 0, 0}, table[2 * 4];
        memset(table, 0, sizeof table);
        memset(&d, 0, sizeof d);
        tabs[0] = table;
        d.num_vars = 1; d.max_multiplicands = 1; d.n_products = 1; d.coeffs = coeff; d.prod_offsets = offs; d.prod_indices = idx;
        d.n_tables = 1; d.tables = tabs; d.flags = 0;
        rc = sc_prover_init(&d, &p);
        if (rc != SC_ERR_HIP || p != NULL || strstr(sc_last_error(), "no CPU fallback") == NULL) {
            printf("expected SC_ERR_HIP without a device, got %d (%s)\n", rc, sc_last_error());
            return 8;
        }
    }
    printf("ABI-SMOKE-OK abi=%d devices=%d\n", sc_abi_version(), sc_device_count());
    return 0;
}
