/* ORACLE (test infrastructure only; never linked into the product library).
 *
 * CPU restatement of the MAC halves of Server::CRebuild_Cached ("FFT in the exponent"), loop for loop:
 *   init scaling   porla/Server/Server.hpp:1523-1536   X[i] = MAC_U[i];  Y[i] = wt * MAC_U[i]
 *   butterflies    Server.hpp:1590-1609, 1658-1676     tm = vi * MAC[k+m2]; MAC[k] = um + tm; MAC[k+m2] = um - tm
 *   stage twiddle  Server.hpp:1553, 1620-1621          v = w^(N/m2) mod p_icc, vi = v^j mod p_icc, handed to the group as
 *                                                      the integer vi (convert_ZZ_to_scalar, utils.h:307-318) and reduced
 *                                                      mod the group order there (fr.SetBytes main.go:209 / secp256k1 scalar)
 * Group law: oracle/curve_a0.h.  The reference's providers (gnark, libsecp256k1, NTL) cannot run here; this file is
 * pinned against the Python restatement oracle/icc_py.py:mac_crebuild (tests/test_oracle_mac.py).  Parity w.r.t. the
 * reference itself is UNPINNED for this path (it holds no tests for it).
 */
#include "curve_a0.h"

static const uint64_t M_BN_P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t M_BN_R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t M_SK_P[4] = {0xFFFFFFFEFFFFFC2Full, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull};
static const uint64_t M_SK_N[4] = {0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull};
static const uint64_t M_P_ICC[4] = {1, 0, 0, 0xcf00000000000000ull};   /* 207 * 2^248 + 1, utils.h:31-32 */
static const uint8_t M_GEN_BE[32] = {0x00,0x15,0x59,0xf5,0x5d,0xe0,0x99,0x9d,0x7b,0x4b,0x58,0xb0,0x69,0x3f,0xe8,0x8e,
                                     0x8d,0xaf,0x5c,0xec,0xc0,0x5b,0x56,0x5b,0x3b,0xd6,0x39,0xda,0x1e,0xa0,0xa8,0xb6};  /* utils.h:29-30 */

typedef struct { curve_t C; mod256 N, PI; u256 w_m; } mac_ctx;

static void mac_init(mac_ctx *X, int curve, size_t n) {
    u256 p, q, pi, g, e, bb = {{curve ? 7 : 3, 0, 0, 0}};
    memcpy(p.l, curve ? M_SK_P : M_BN_P, 32); memcpy(q.l, curve ? M_SK_N : M_BN_R, 32); memcpy(pi.l, M_P_ICC, 32);
    mod256_init(&X->C.F, &p); mod256_init(&X->N, &q); mod256_init(&X->PI, &pi);
    mod_to_mont(&X->C.F, &X->C.b, &bb);
    /* w = GENERATOR^((p_icc-1)/(2N)) mod p_icc, Server.hpp:214-216 */
    u256_from_be(&g, M_GEN_BE);
    e = pi; e.l[0] -= 1;
    int sh = 1; while (((size_t)1 << (sh - 1)) < n) sh++;
    for (int s = 0; s < sh; s++) { for (int i = 0; i < 3; i++) e.l[i] = (e.l[i] >> 1) | (e.l[i + 1] << 63); e.l[3] >>= 1; }
    u256 gm; mod_to_mont(&X->PI, &gm, &g);
    mod_pow(&X->PI, &X->w_m, &gm, &e);
}
static void mac_in(const mac_ctx *X, aff_t *a, const uint8_t b[64]) {
    u256 x, y; u256_from_be(&x, b); u256_from_be(&y, b + 32);
    mod256_reduce(&X->C.F, &x); mod256_reduce(&X->C.F, &y);
    if (u256_is_zero(&x) && u256_is_zero(&y)) { memset(a, 0, sizeof(*a)); a->inf = 1; return; }
    mod_to_mont(&X->C.F, &a->x, &x); mod_to_mont(&X->C.F, &a->y, &y); a->inf = 0;
}
static void mac_out(const mac_ctx *X, uint8_t b[64], const jac_t *p) {
    aff_t a; jac_to_aff(&X->C, &a, p);
    if (a.inf) { memset(b, 0, 64); return; }
    u256 x, y; mod_from_mont(&X->C.F, &x, &a.x); mod_from_mont(&X->C.F, &y, &a.y);
    u256_to_be(b, &x); u256_to_be(b + 32, &y);
}
static uint64_t mac_rev_bits(uint64_t x, int n) { uint64_t r = 0; for (int i = 0; i < n; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
/* k * P with k = (integer value of the p_icc residue vi) mod group order */
static void mac_mul(const mac_ctx *X, jac_t *r, const jac_t *p, const u256 *vi_plain) {
    u256 k = *vi_plain; mod256_reduce(&X->N, &k);
    jac_t acc; jac_set_inf(&X->C, &acc);
    for (int i = 255; i >= 0; i--) { jac_double(&X->C, &acc, &acc); if (u256_bit(&k, i)) jac_add(&X->C, &acc, &acc, p); }
    *r = acc;
}

/* macs_in / macs_out: n points, 64-byte X||Y big-endian affine (zeros = infinity); is_y selects the Y part */
void oracle_icc_mac_crebuild(const uint8_t *macs_in, size_t n, int curve, int is_y, uint64_t write_step, uint8_t *macs_out,
                             int threads) {
    mac_ctx X; mac_init(&X, curve, n);
    int height = 1; while (((size_t)1 << (height - 1)) < n) height++;   /* ceil(log2 n) + 1, Server.hpp:219 */
    jac_t *M = (jac_t *)malloc(sizeof(jac_t) * n);
    u256 wt_m, wt, ee = {{is_y ? mac_rev_bits(write_step % n, height - 1) : 0, 0, 0, 0}};
    mod_pow(&X.PI, &wt_m, &X.w_m, &ee); mod_from_mont(&X.PI, &wt, &wt_m);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < n; i++) {
        aff_t a; mac_in(&X, &a, macs_in + 64 * i);
        jac_from_aff(&X.C, &M[i], &a);
        if (is_y) { jac_t t; mac_mul(&X, &t, &M[i], &wt); M[i] = t; }
    }
    for (int s = 1; s < height; s++) {
        size_t m = (size_t)1 << s, m2 = m >> 1;
        u256 v_m, e2 = {{n / m2, 0, 0, 0}}; mod_pow(&X.PI, &v_m, &X.w_m, &e2);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
        for (size_t j = 0; j < m2; j++) {
            u256 vi_m, vi, ej = {{j, 0, 0, 0}};
            mod_pow(&X.PI, &vi_m, &v_m, &ej); mod_from_mont(&X.PI, &vi, &vi_m);    /* vi = v^j (Server.hpp:1635) */
            for (size_t k = j; k < n; k += m) {
                jac_t tm, um = M[k], ntm;
                mac_mul(&X, &tm, &M[k + m2], &vi);
                jac_add(&X.C, &M[k], &um, &tm);
                jac_neg(&X.C, &ntm, &tm);
                jac_add(&X.C, &M[k + m2], &um, &ntm);
            }
        }
    }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < n; i++) mac_out(&X, macs_out + 64 * i, &M[i]);
    free(M);
}

/* MAC part of Server::mix (Server.hpp:1281-1318; the same lines serve the alignment MACs :1300-1318):
 * out[i] = A0[i] + v^i * A1[i], out[i+len] = A0[i] - v^i * A1[i], v = w^(n_total/len) with w of n_total blocks */
void oracle_icc_mac_mix(const uint8_t *a0, const uint8_t *a1, size_t len, size_t n_total, int curve, uint8_t *out, int threads) {
    mac_ctx X; mac_init(&X, curve, n_total);
    u256 v_m, e = {{n_total / len, 0, 0, 0}}; mod_pow(&X.PI, &v_m, &X.w_m, &e);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (size_t i = 0; i < len; i++) {
        u256 vi_m, vi, ei = {{i, 0, 0, 0}};
        mod_pow(&X.PI, &vi_m, &v_m, &ei); mod_from_mont(&X.PI, &vi, &vi_m);
        aff_t p0, p1; mac_in(&X, &p0, a0 + 64 * i); mac_in(&X, &p1, a1 + 64 * i);
        jac_t j0, j1, tm, ntm, r;
        jac_from_aff(&X.C, &j0, &p0); jac_from_aff(&X.C, &j1, &p1);
        mac_mul(&X, &tm, &j1, &vi);
        jac_add(&X.C, &r, &j0, &tm); mac_out(&X, out + 64 * i, &r);
        jac_neg(&X.C, &ntm, &tm);
        jac_add(&X.C, &r, &j0, &ntm); mac_out(&X, out + 64 * (i + len), &r);
    }
}
