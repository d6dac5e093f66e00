/* ORACLE (test infrastructure only; never linked into the product library).
 *
 * CPU restatement of the IPA back-end's group arithmetic: what Porla obtains from the vendored
 * libsecp256k1 internals through secp256k1_ecmult_multi_var (porla/Utils/secp256k1_lib/ecmult_impl.h:814-860,
 * call sites porla/Client/Client.hpp:395,778, porla/Server/Server.hpp:349,842-848) and secp256k1_ecmult
 * (ecmult_impl.h:335-349).  The reference algorithm (GLV split :621-634, fixed-window wNAF :413-473, bucket
 * accumulation :492-567, Strauss :220-333) computes R = sum s_i * P_i as a group element; parity is defined on
 * the normalised affine point, so this file restates the group law (oracle/curve_a0.h; the
 * reference's gej_double / gej_add_var / gej_add_ge_var are group_impl.h:274-306, :336-387, :389-436) and the
 * plain definition of the sum.
 *
 * The vendored tree cannot be compiled here without writing a stand-in for the absent public header
 * include/secp256k1.h (secp256k1.c:9-10), so there is no oracle/_ref build.  PINNED instead against the
 * reference's own known-answer tests, restated in tests/test_oracle_secp256k1.py:
 *   - test_ecmult_constants   tests.c:4715-4757  (SHA-256 over 32 842 serialised x*G, expected hash e4711b4d...)
 *   - run_ecmult_chain        tests.c:3493-3555  (point after 20 000 iterations of X = xn*X + gn*G)
 * Constants: p field_5x52.h:13-15, n scalar_4x64_impl.h:13-16, G group_impl.h:28-33, b = 7 group_impl.h:64.
 */
#include "curve_a0.h"

static curve_t SK;
static mod256 SKN;
static aff_t SK_G;
static int sk_ready = 0;

static const uint64_t SK_P[4] = {0xFFFFFFFEFFFFFC2Full, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull};
static const uint64_t SK_N[4] = {0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull};
static const uint8_t SK_GX[32] = {0x79,0xBE,0x66,0x7E,0xF9,0xDC,0xBB,0xAC,0x55,0xA0,0x62,0x95,0xCE,0x87,0x0B,0x07,
                                  0x02,0x9B,0xFC,0xDB,0x2D,0xCE,0x28,0xD9,0x59,0xF2,0x81,0x5B,0x16,0xF8,0x17,0x98};
static const uint8_t SK_GY[32] = {0x48,0x3A,0xDA,0x77,0x26,0xA3,0xC4,0x65,0x5D,0xA4,0xFB,0xFC,0x0E,0x11,0x08,0xA8,
                                  0xFD,0x17,0xB4,0x48,0xA6,0x85,0x54,0x19,0x9C,0x47,0xD0,0x8F,0xFB,0x10,0xD4,0xB8};

static void sk_init(void) {
    if (sk_ready) return;
    u256 p, n, x, y, seven = {{7, 0, 0, 0}};
    memcpy(p.l, SK_P, 32); memcpy(n.l, SK_N, 32);
    mod256_init(&SK.F, &p); mod256_init(&SKN, &n);
    mod_to_mont(&SK.F, &SK.b, &seven);
    u256_from_be(&x, SK_GX); u256_from_be(&y, SK_GY);
    mod_to_mont(&SK.F, &SK_G.x, &x); mod_to_mont(&SK.F, &SK_G.y, &y); SK_G.inf = 0;
    sk_ready = 1;
}
/* canonical encodings at the C ABI of the engine: 32-byte BE scalar (reduced mod n), 64-byte x||y BE, zeros = infinity */
static void sk_point_in(aff_t *a, const uint8_t b[64]) {
    u256 x, y; u256_from_be(&x, b); u256_from_be(&y, b + 32);
    mod256_reduce(&SK.F, &x); mod256_reduce(&SK.F, &y);
    if (u256_is_zero(&x) && u256_is_zero(&y)) { memset(a, 0, sizeof(*a)); a->inf = 1; return; }
    mod_to_mont(&SK.F, &a->x, &x); mod_to_mont(&SK.F, &a->y, &y); a->inf = 0;
}
static void sk_point_out(uint8_t b[64], const jac_t *p) {
    aff_t a; jac_to_aff(&SK, &a, p);
    if (a.inf) { memset(b, 0, 64); return; }
    u256 x, y; mod_from_mont(&SK.F, &x, &a.x); mod_from_mont(&SK.F, &y, &a.y);
    u256_to_be(b, &x); u256_to_be(b + 32, &y);
}
static void sk_scalar_in(u256 *k, const uint8_t b[32]) { u256_from_be(k, b); mod256_reduce(&SKN, k); }

void oracle_secp256k1_generator(uint8_t out[64]) { sk_init(); jac_t j; jac_from_aff(&SK, &j, &SK_G); sk_point_out(out, &j); }

/* R = sum s_i * P_i  (secp256k1_ecmult_multi_var with inp_g_sc = 0, ecmult_impl.h:814-860) */
void oracle_secp256k1_multi(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[64], int threads, int naive) {
    sk_init();
    u256 *k = (u256 *)malloc(sizeof(u256) * (n ? n : 1));
    aff_t *pts = (aff_t *)malloc(sizeof(aff_t) * (n ? n : 1));
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < n; i++) { sk_scalar_in(&k[i], scalars + 32 * i); sk_point_in(&pts[i], points + 64 * i); }
    jac_t r;
    if (naive) msm_naive(&SK, &r, k, pts, n); else msm_pippenger(&SK, &r, k, pts, n, 256, threads);
    sk_point_out(out, &r);
    free(k); free(pts);
}
/* Pedersen commitment row by row (compute_commitment, Client.hpp:374-406 / Server.hpp:329-361: ecmult_multi_var of the
 * 128 chunks of a block against the fixed generators): out[r] = sum_i row_r[i] * base[i] */
void oracle_secp256k1_commit_batch(const uint8_t *rows, size_t n_rows, size_t n_coeffs, size_t row_stride,
                                   const uint8_t *base, uint8_t *out, int threads, int naive) {
    sk_init();
    aff_t *pts = (aff_t *)malloc(sizeof(aff_t) * (n_coeffs ? n_coeffs : 1));
    for (size_t i = 0; i < n_coeffs; i++) sk_point_in(&pts[i], base + 64 * i);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
    for (size_t r = 0; r < n_rows; r++) {
        u256 *k = (u256 *)malloc(sizeof(u256) * (n_coeffs ? n_coeffs : 1));
        for (size_t i = 0; i < n_coeffs; i++) sk_scalar_in(&k[i], rows + r * row_stride + 32 * i);
        jac_t j;
        if (naive) msm_naive(&SK, &j, k, pts, n_coeffs); else msm_pippenger(&SK, &j, k, pts, n_coeffs, 256, 1);
        sk_point_out(out + 64 * r, &j);
        free(k);
    }
    free(pts);
}
/* R = na * A + ng * G  (secp256k1_ecmult, ecmult_impl.h:335-349) */
void oracle_secp256k1_ecmult(const uint8_t a[64], const uint8_t na[32], const uint8_t ng[32], uint8_t out[64]) {
    sk_init();
    aff_t pa; u256 ka, kg; jac_t r1, r2;
    sk_point_in(&pa, a); sk_scalar_in(&ka, na); sk_scalar_in(&kg, ng);
    jac_mul_aff(&SK, &r1, &pa, &ka); jac_mul_aff(&SK, &r2, &SK_G, &kg); jac_add(&SK, &r1, &r1, &r2);
    sk_point_out(out, &r1);
}
int oracle_secp256k1_on_curve(const uint8_t a[64]) { sk_init(); aff_t pa; sk_point_in(&pa, a); return aff_on_curve(&SK, &pa); }

/* run_ecmult_chain, tests.c:3493-3555: X <- xn*X + gn*G; xn *= xf; gn *= gf, `iters` times; returns X */
void oracle_secp256k1_ecmult_chain(const uint8_t a[64], const uint8_t xn0[32], const uint8_t gn0[32], uint32_t xf, uint32_t gf,
                                   int iters, uint8_t out[64]) {
    sk_init();
    aff_t X; sk_point_in(&X, a);
    u256 xn, gn, xnm, gnm, xfm, gfm, t;
    sk_scalar_in(&xn, xn0); sk_scalar_in(&gn, gn0);
    u256 f1 = {{xf, 0, 0, 0}}, f2 = {{gf, 0, 0, 0}};
    mod_to_mont(&SKN, &xnm, &xn); mod_to_mont(&SKN, &gnm, &gn); mod_to_mont(&SKN, &xfm, &f1); mod_to_mont(&SKN, &gfm, &f2);
    jac_t r1, r2;
    for (int i = 0; i < iters; i++) {
        mod_from_mont(&SKN, &t, &xnm); jac_mul_aff(&SK, &r1, &X, &t);
        mod_from_mont(&SKN, &t, &gnm); jac_mul_aff(&SK, &r2, &SK_G, &t);
        jac_add(&SK, &r1, &r1, &r2);
        jac_to_aff(&SK, &X, &r1);
        mod_mul(&SKN, &xnm, &xnm, &xfm); mod_mul(&SKN, &gnm, &gnm, &gfm);
    }
    jac_t j; jac_from_aff(&SK, &j, &X); sk_point_out(out, &j);
}
/* x*G for a batch of 32-byte BE scalars (test_ecmult_accumulate, tests.c:4681-4713, serialisation left to the caller) */
void oracle_secp256k1_mul_g_batch(const uint8_t *k_be, size_t n, uint8_t *out, int threads) {
    sk_init();
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 64)
    for (size_t i = 0; i < n; i++) {
        u256 k; sk_scalar_in(&k, k_be + 32 * i);
        jac_t j; jac_mul_aff(&SK, &j, &SK_G, &k); sk_point_out(out + 64 * i, &j);
    }
}
/* bench_ecmult.c:328-337 generator: P_i = 2^i * G, normalised; i in [0, n) */
void oracle_secp256k1_doubling_chain(size_t n, uint8_t *out) {
    sk_init();
    const size_t CH = 4096;
    jac_t cur; jac_from_aff(&SK, &cur, &SK_G);
    jac_t *buf = (jac_t *)malloc(sizeof(jac_t) * CH);
    aff_t *af = (aff_t *)malloc(sizeof(aff_t) * CH);
    for (size_t lo = 0; lo < n; lo += CH) {
        size_t m = n - lo < CH ? n - lo : CH;
        for (size_t i = 0; i < m; i++) { buf[i] = cur; jac_double(&SK, &cur, &cur); }
        jac_batch_to_aff(&SK, af, buf, m);
        for (size_t i = 0; i < m; i++) {
            u256 x, y; mod_from_mont(&SK.F, &x, &af[i].x); mod_from_mont(&SK.F, &y, &af[i].y);
            u256_to_be(out + 64 * (lo + i), &x); u256_to_be(out + 64 * (lo + i) + 32, &y);
        }
    }
    free(buf); free(af);
}
