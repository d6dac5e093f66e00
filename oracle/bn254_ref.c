/* ORACLE (test infrastructure only; never linked into the product library).
 *
 * CPU restatement of the KZG/BN254 plug-in of the reference: every exported function below
 * restates one cgo export of porla/main.go (file:line cited per function).  The arithmetic
 * provider of the reference, gnark-crypto v0.6.0, is not under /root/reference and cannot be
 * fetched or built here (no Go, no network): the group law is restated from the public
 * definition of BN254 (oracle/curve_a0.h) and the byte formats from gnark's published
 * marshal format (see oracle/bn254_py.py header).  PARITY w.r.t. gnark itself is UNPINNED
 * (the reference holds no tests for this path); this file is pinned against
 * oracle/bn254_py.py (independent Python big-int code), EIP-196 known answers and the
 * in-reference identities listed there -- see tests/test_oracle_bn254.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared).
 */
#include "curve_a0.h"
#include <stdio.h>

static curve_t BN;           /* base field Fp + b = 3 */
static mod256 FR;            /* scalar field */
static aff_t BN_G;           /* (1, 2) */
static int bn_ready = 0;

static const uint64_t BN_P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t BN_R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

static void bn_init(void) {
    if (bn_ready) return;
    u256 p, r;
    memcpy(p.l, BN_P, 32); memcpy(r.l, BN_R, 32);
    mod256_init(&BN.F, &p);
    mod256_init(&FR, &r);
    u256 three = {{3, 0, 0, 0}}, one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    mod_to_mont(&BN.F, &BN.b, &three);
    mod_to_mont(&BN.F, &BN_G.x, &one); mod_to_mont(&BN.F, &BN_G.y, &two); BN_G.inf = 0;
    bn_ready = 1;
}

/* fr.Element.SetBytes (main.go:127): 32-byte big-endian, reduced mod r; returned as a plain integer */
static void fr_set_bytes(u256 *k, const uint8_t b[32]) { u256_from_be(k, b); mod256_reduce(&FR, k); }

static int fp_sqrt(u256 *r, const u256 *a) { /* p = 3 mod 4 */
    u256 e = BN.F.m, one = {{1, 0, 0, 0}}, s, chk;
    u256_add(&e, &e, &one);
    for (int i = 0; i < 3; i++) e.l[i] = (e.l[i] >> 2) | (e.l[i + 1] << 62);
    e.l[3] >>= 2;
    mod_pow(&BN.F, &s, a, &e);
    mod_sqr(&BN.F, &chk, &s);
    *r = s;
    return u256_eq(&chk, a);
}
static int fp_lex_largest(const u256 *a_mont) { /* y > (p-1)/2 on the regular value */
    u256 v, half = BN.F.m;
    mod_from_mont(&BN.F, &v, a_mont);
    for (int i = 0; i < 3; i++) half.l[i] = (half.l[i] >> 1) | (half.l[i + 1] << 63);
    half.l[3] >>= 1;
    return !u256_geq(&half, &v);
}

/* G1Affine.Unmarshal of a 64-byte buffer (main.go:130): flags 00 -> X,Y = SetBytes (mod p);
 * (0,0) = infinity; other flag values are gnark's compressed forms */
static void g1_unmarshal(aff_t *a, const uint8_t b[64]) {
    uint8_t flags = b[0] & 0xC0;
    u256 x, y;
    if (flags == 0x00) {
        u256_from_be(&x, b); u256_from_be(&y, b + 32);
        mod256_reduce(&BN.F, &x); mod256_reduce(&BN.F, &y);
        if (u256_is_zero(&x) && u256_is_zero(&y)) { memset(a, 0, sizeof(*a)); a->inf = 1; return; }
        mod_to_mont(&BN.F, &a->x, &x); mod_to_mont(&BN.F, &a->y, &y); a->inf = 0;
        return;
    }
    if (flags == 0x40) { memset(a, 0, sizeof(*a)); a->inf = 1; return; }
    uint8_t t[32]; memcpy(t, b, 32); t[0] &= 0x3F;
    u256_from_be(&x, t); mod256_reduce(&BN.F, &x);
    mod_to_mont(&BN.F, &a->x, &x);
    u256 rhs; mod_sqr(&BN.F, &rhs, &a->x); mod_mul(&BN.F, &rhs, &rhs, &a->x); mod_add(&BN.F, &rhs, &rhs, &BN.b);
    fp_sqrt(&y, &rhs);
    if (fp_lex_largest(&y) != (flags == 0xC0)) mod_neg(&BN.F, &y, &y);
    a->y = y; a->inf = 0;
}
/* G1Affine.Marshal (main.go:137): X||Y big-endian regular form; infinity = 64 zero bytes */
static void g1_marshal(uint8_t b[64], const aff_t *a) {
    if (a->inf) { memset(b, 0, 64); return; }
    u256 x, y;
    mod_from_mont(&BN.F, &x, &a->x); mod_from_mont(&BN.F, &y, &a->y);
    u256_to_be(b, &x); u256_to_be(b + 32, &y);
}
static void g1_marshal_jac(uint8_t b[64], const jac_t *p) { aff_t a; jac_to_aff(&BN, &a, p); g1_marshal(b, &a); }
static void g1_compress(uint8_t b[32], const aff_t *a) {
    if (a->inf) { memset(b, 0, 32); b[0] = 0x40; return; }
    u256 x; mod_from_mont(&BN.F, &x, &a->x); u256_to_be(b, &x);
    b[0] |= fp_lex_largest(&a->y) ? 0xC0 : 0x80;
}

/* ------------------------------------------------------------------ group entry points */
/* compute_multi_exp, main.go:118-138.  naive != 0 -> sum of double-and-add products */
void oracle_bn254_multi_exp(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[64],
                            int threads, int naive) {
    bn_init();
    u256 *k = (u256 *)malloc(sizeof(u256) * (n ? n : 1));
    aff_t *pts = (aff_t *)malloc(sizeof(aff_t) * (n ? n : 1));
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < n; i++) { fr_set_bytes(&k[i], scalars + 32 * i); g1_unmarshal(&pts[i], points + 64 * i); }
    jac_t r;
    if (naive) msm_naive(&BN, &r, k, pts, n); else msm_pippenger(&BN, &r, k, pts, n, 254, threads);
    g1_marshal_jac(out, &r);
    free(k); free(pts);
}
/* kzg.Commit row by row (compute_digest_from_srs, main.go:103-116, called per row at Server.hpp:550-560): out[r] =
 * sum_i (row_r[i] mod r) * base[i]; one MSM per row (naive != 0: double-and-add sum), rows split over threads */
void oracle_bn254_commit_batch(const uint8_t *rows, size_t n_rows, size_t n_coeffs, size_t row_stride,
                               const uint8_t *base, uint8_t *out, int threads, int naive) {
    bn_init();
    aff_t *pts = (aff_t *)malloc(sizeof(aff_t) * (n_coeffs ? n_coeffs : 1));
    for (size_t i = 0; i < n_coeffs; i++) g1_unmarshal(&pts[i], base + 64 * i);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
    for (size_t r = 0; r < n_rows; r++) {
        u256 *k = (u256 *)malloc(sizeof(u256) * (n_coeffs ? n_coeffs : 1));
        for (size_t i = 0; i < n_coeffs; i++) fr_set_bytes(&k[i], rows + r * row_stride + 32 * i);
        jac_t j;
        if (naive) msm_naive(&BN, &j, k, pts, n_coeffs); else msm_pippenger(&BN, &j, k, pts, n_coeffs, 254, 1);
        g1_marshal_jac(out + 64 * r, &j);
        free(k);
    }
    free(pts);
}
/* add_point, main.go:195-202 (in place on a) */
void oracle_bn254_add_point(uint8_t a[64], const uint8_t b[64]) {
    bn_init();
    aff_t pa, pb; jac_t j;
    g1_unmarshal(&pa, a); g1_unmarshal(&pb, b);
    jac_from_aff(&BN, &j, &pa); jac_add_aff(&BN, &j, &j, &pb);
    g1_marshal_jac(a, &j);
}
/* mult_point, main.go:204-214 (scalar SetBytes-reduced, in place) */
void oracle_bn254_mult_point(uint8_t a[64], const uint8_t s[32]) {
    bn_init();
    aff_t pa; u256 k; jac_t j;
    g1_unmarshal(&pa, a); fr_set_bytes(&k, s);
    jac_mul_aff(&BN, &j, &pa, &k);
    g1_marshal_jac(a, &j);
}
/* neg_point, main.go:216-222 */
void oracle_bn254_neg_point(uint8_t a[64]) {
    bn_init();
    aff_t pa; g1_unmarshal(&pa, a);
    if (!pa.inf) mod_neg(&BN.F, &pa.y, &pa.y);
    g1_marshal(a, &pa);
}
/* compare_commitment, main.go:140-151 */
int oracle_bn254_compare(const uint8_t a[64], const uint8_t b[64]) {
    bn_init();
    aff_t pa, pb; g1_unmarshal(&pa, a); g1_unmarshal(&pb, b);
    if (pa.inf || pb.inf) return pa.inf == pb.inf;
    return u256_eq(&pa.x, &pb.x) && u256_eq(&pa.y, &pb.y);
}
int oracle_bn254_on_curve(const uint8_t a[64]) {
    bn_init();
    aff_t pa; g1_unmarshal(&pa, a);
    return aff_on_curve(&BN, &pa);
}

/* synthetic inputs: out[i] = k_i * G for plain 32-byte BE k_i, fixed-base 8-bit windows */
void oracle_bn254_fixed_base(const uint8_t *k_be, size_t n, uint8_t *out, int threads) {
    bn_init();
    /* table[w][d-1] = d * 2^(8w) * G, affine */
    enum { W = 32, D = 255 };
    aff_t *table = (aff_t *)malloc(sizeof(aff_t) * W * D);
    jac_t *tj = (jac_t *)malloc(sizeof(jac_t) * W * D);
    jac_t base; jac_from_aff(&BN, &base, &BN_G);
    for (int w = 0; w < W; w++) {
        jac_t acc = base;
        for (int d = 0; d < D; d++) { tj[w * D + d] = acc; jac_add(&BN, &acc, &acc, &base); }
        base = acc; /* 256 * base */
    }
    jac_batch_to_aff(&BN, table, tj, (size_t)W * D);
    free(tj);
    if (threads < 1) threads = 1;
    const size_t CH = 4096;
    size_t nch = (n + CH - 1) / CH;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (size_t ch = 0; ch < nch; ch++) {
        size_t lo = ch * CH, hi = lo + CH > n ? n : lo + CH;
        jac_t *acc = (jac_t *)malloc(sizeof(jac_t) * (hi - lo));
        aff_t *af = (aff_t *)malloc(sizeof(aff_t) * (hi - lo));
        for (size_t i = lo; i < hi; i++) {
            u256 k; u256_from_be(&k, k_be + 32 * i); mod256_reduce(&FR, &k);
            jac_t a; jac_set_inf(&BN, &a);
            for (int w = 0; w < W; w++) {
                uint32_t d = u256_bits(&k, 8 * w, 8);
                if (d) jac_add_aff(&BN, &a, &a, &table[w * D + d - 1]);
            }
            acc[i - lo] = a;
        }
        jac_batch_to_aff(&BN, af, acc, hi - lo);
        for (size_t i = lo; i < hi; i++) g1_marshal(out + 64 * i, &af[i - lo]);
        free(acc); free(af);
    }
    free(table);
}

/* ------------------------------------------------------------------ KZG state, main.go:18-29 */
static u256 kz_tau, kz_alpha;         /* plain integers mod r */
static u256 kz_tau_bi;                /* big.Int of the raw key bytes (main.go:35) */
static size_t kz_n = 0;
static aff_t *kz_srs = NULL;
static aff_t kz_hmac;

static void be_var(u256 *r, const uint8_t *b, size_t len) { /* big-endian, len <= 32 */
    uint8_t t[32]; memset(t, 0, 32);
    if (len > 32) { b += len - 32; len = 32; }
    memcpy(t + 32 - len, b, len);
    u256_from_be(r, t);
}
/* init_key, main.go:31-40 */
void oracle_kzg_init_key(const uint8_t *tau, size_t tau_len, const uint8_t *alpha, size_t alpha_len) {
    bn_init();
    be_var(&kz_tau_bi, tau, tau_len); kz_tau = kz_tau_bi; mod256_reduce(&FR, &kz_tau);
    be_var(&kz_alpha, alpha, alpha_len); mod256_reduce(&FR, &kz_alpha);
}
/* init_SRS, main.go:42-60: G1[i] = tau^i * G; h_MAC = h_scalar * G1[0] (the reference draws h_scalar at random) */
void oracle_kzg_init_srs(size_t n, const uint8_t h_scalar[32]) {
    bn_init();
    free(kz_srs);
    kz_n = n; kz_srs = (aff_t *)malloc(sizeof(aff_t) * (n ? n : 1));
    u256 t = {{1, 0, 0, 0}}, tm, taum;
    mod_to_mont(&FR, &taum, &kz_tau);
    mod_to_mont(&FR, &tm, &t);
    for (size_t i = 0; i < n; i++) {
        u256 e; mod_from_mont(&FR, &e, &tm);
        jac_t j; jac_mul_aff(&BN, &j, &BN_G, &e); jac_to_aff(&BN, &kz_srs[i], &j);
        mod_mul(&FR, &tm, &tm, &taum);
    }
    u256 h; fr_set_bytes(&h, h_scalar);
    jac_t j; jac_mul_aff(&BN, &j, &kz_srs[0], &h); jac_to_aff(&BN, &kz_hmac, &j);
}
/* G1 part of SRS.WriteTo (main.go:48): 4-byte BE count || n x 32-byte compressed */
void oracle_kzg_srs_g1_blob(uint8_t *out) {
    out[0] = (uint8_t)(kz_n >> 24); out[1] = (uint8_t)(kz_n >> 16); out[2] = (uint8_t)(kz_n >> 8); out[3] = (uint8_t)kz_n;
    for (size_t i = 0; i < kz_n; i++) g1_compress(out + 4 + 32 * i, &kz_srs[i]);
}
void oracle_kzg_srs_g1_raw(uint8_t *out) { for (size_t i = 0; i < kz_n; i++) g1_marshal(out + 64 * i, &kz_srs[i]); }

static void poly_load(u256 *f, const uint8_t *data) { for (size_t i = 0; i < kz_n; i++) fr_set_bytes(&f[i], data + 32 * i); }
static void fr_horner(u256 *y, const u256 *f, size_t n, const u256 *z) { /* plain in, plain out */
    u256 acc, zm, cm; memset(&acc, 0, sizeof(acc));
    mod_to_mont(&FR, &zm, z);
    for (size_t i = n; i-- > 0;) { mod_mul(&FR, &acc, &acc, &zm); mod_to_mont(&FR, &cm, &f[i]); mod_add(&FR, &acc, &acc, &cm); }
    mod_from_mont(&FR, y, &acc);
}
/* compute_digest, main.go:70-89: alpha * f(tau) * G1[0] */
void oracle_kzg_compute_digest(const uint8_t *data, uint8_t out[64]) {
    u256 *f = (u256 *)malloc(sizeof(u256) * kz_n); poly_load(f, data);
    u256 fx, fm, am; fr_horner(&fx, f, kz_n, &kz_tau);
    mod_to_mont(&FR, &fm, &fx); mod_to_mont(&FR, &am, &kz_alpha); mod_mul(&FR, &fm, &fm, &am); mod_from_mont(&FR, &fx, &fm);
    jac_t j; jac_mul_aff(&BN, &j, &kz_srs[0], &fx); g1_marshal_jac(out, &j);
    free(f);
}
/* compute_digest_complement, main.go:91-101 */
void oracle_kzg_compute_digest_complement(const uint8_t data[32], uint8_t out[64]) {
    u256 k; fr_set_bytes(&k, data);
    jac_t j; jac_mul_aff(&BN, &j, &kz_hmac, &k); g1_marshal_jac(out, &j);
}
/* compute_digest_from_srs, main.go:103-116: kzg.Commit = MSM(coefficients, SRS.G1) */
void oracle_kzg_compute_digest_from_srs(const uint8_t *data, uint8_t out[64]) {
    u256 *f = (u256 *)malloc(sizeof(u256) * kz_n); poly_load(f, data);
    jac_t j; msm_pippenger(&BN, &j, f, kz_srs, kz_n, 254, 1); g1_marshal_jac(out, &j);
    free(f);
}
/* create_proof, main.go:153-175 (kzg.Commit + kzg.Open) */
void oracle_kzg_create_proof(uint64_t random_point, const uint8_t *data, uint8_t commitment[64],
                             uint8_t proof_h[64], uint8_t proof_point[32], uint8_t proof_claim[32]) {
    u256 *f = (u256 *)malloc(sizeof(u256) * kz_n); poly_load(f, data);
    jac_t j; msm_pippenger(&BN, &j, f, kz_srs, kz_n, 254, 1); g1_marshal_jac(commitment, &j);
    u256 z = {{random_point, 0, 0, 0}}, y;
    fr_horner(&y, f, kz_n, &z);
    /* h = (f - y) / (X - z): h[n-2] = f[n-1], h[i-1] = f[i] + z*h[i] */
    u256 *h = (u256 *)malloc(sizeof(u256) * kz_n);
    u256 zm, carry, cm; mod_to_mont(&FR, &zm, &z); memset(&carry, 0, sizeof(carry));
    for (size_t i = kz_n - 1; i >= 1; i--) {
        mod_mul(&FR, &carry, &carry, &zm); mod_to_mont(&FR, &cm, &f[i]); mod_add(&FR, &carry, &carry, &cm);
        mod_from_mont(&FR, &h[i - 1], &carry);
    }
    msm_pippenger(&BN, &j, h, kz_srs, kz_n - 1, 254, 1); g1_marshal_jac(proof_h, &j);
    u256_to_be(proof_point, &z); u256_to_be(proof_claim, &y);
    free(f); free(h);
}
