/* ORACLE (test infrastructure only; never linked into the product library).
 *
 * C restatement of the ICC encode arithmetic of the reference (NTL ZZ / ZZ_p in the original; NTL is absent
 * here, so the reference path cannot run and holds no tests -> pinned against oracle/icc_py.py only):
 *   CRebuild_Cached X/Y parts   porla/Server/Server.hpp:1548-1687, 1691-1830 (init scaling :1494,1512-1522)
 *   mix                          porla/Server/Server.hpp:1269-1278
 *   align_MAC scalar part        porla/Server/Server.hpp:531-541 (KZG), :495-504 (IPA)
 *   constants                    porla/Utils/utils.h:27-43
 * Arithmetic is done DIRECTLY in Z/(LCM), LCM = p_icc * q (510/512 bits), with an 8 x 64-bit Montgomery
 * multiplier -- deliberately a different method from the engine's (p_icc, q) residue pair, so that agreement
 * between the two is meaningful.  The loop nest, twiddle recurrence (vi *= v) and the "% LCM" reductions follow
 * the reference line for line.  Formats: elements in = 32-byte little-endian (utils.h:353-364), rows out =
 * 64-byte little-endian (utils.h:473-517), alignment scalars = 32-byte big-endian (utils.h:307-318).
 */
#include "mont256.h"
#include <stdlib.h>

typedef struct { uint64_t l[8]; } u512;
typedef struct { u512 m; uint64_t inv; u512 r2; } mod512;

static int u512_geq(const u512 *a, const u512 *b) {
    for (int i = 7; i >= 0; i--) if (a->l[i] != b->l[i]) return a->l[i] > b->l[i];
    return 1;
}
static uint64_t u512_add(u512 *r, const u512 *a, const u512 *b) {
    u128 c = 0;
    for (int i = 0; i < 8; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t u512_sub(u512 *r, const u512 *a, const u512 *b) {
    uint64_t br = 0;
    for (int i = 0; i < 8; i++) { u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    return br;
}
static void m512_add(const mod512 *M, u512 *r, const u512 *a, const u512 *b) {
    uint64_t c = u512_add(r, a, b);
    if (c || u512_geq(r, &M->m)) u512_sub(r, r, &M->m);
}
static void m512_sub(const mod512 *M, u512 *r, const u512 *a, const u512 *b) {
    if (u512_sub(r, a, b)) u512_add(r, r, &M->m);
}
static void m512_mul(const mod512 *M, u512 *r, const u512 *a, const u512 *b) {
    uint64_t t[10] = {0};
    for (int i = 0; i < 8; i++) {
        u128 c = 0;
        for (int j = 0; j < 8; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[8]; t[8] = (uint64_t)c; t[9] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * M->inv;
        c = (u128)q * M->m.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 8; j++) { c += (u128)q * M->m.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[8]; t[7] = (uint64_t)c; t[8] = t[9] + (uint64_t)(c >> 64);
    }
    u512 x; memcpy(x.l, t, 64);
    if (t[8] || u512_geq(&x, &M->m)) u512_sub(&x, &x, &M->m);
    *r = x;
}
static void m512_init(mod512 *M, const u512 *m) {
    M->m = *m;
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - m->l[0] * x;
    M->inv = (uint64_t)0 - x;
    u512 v; memset(&v, 0, sizeof v); v.l[0] = 1;
    for (int i = 0; i < 1024; i++) { uint64_t c = u512_add(&v, &v, &v); if (c || u512_geq(&v, m)) u512_sub(&v, &v, m); }
    M->r2 = v;
}
static void m512_to(const mod512 *M, u512 *r, const u512 *a) { m512_mul(M, r, a, &M->r2); }
static void m512_from(const mod512 *M, u512 *r, const u512 *a) { u512 one; memset(&one, 0, sizeof one); one.l[0] = 1; m512_mul(M, r, a, &one); }

/* ---------------------------------------------------------------- constants, utils.h:27-43 */
static const uint64_t P_ICC[4] = {1, 0, 0, 0xcf00000000000000ull};
static const uint8_t GEN_BE[32] = {0x00,0x15,0x59,0xf5,0x5d,0xe0,0x99,0x9d,0x7b,0x4b,0x58,0xb0,0x69,0x3f,0xe8,0x8e,0x8d,0xaf,0x5c,0xec,0xc0,0x5b,0x56,0x5b,0x3b,0xd6,0x39,0xda,0x1e,0xa0,0xa8,0xb6};  /* GENERATOR, utils.h:29-30 */
static const uint64_t Q_BN[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t Q_SK[4] = {0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull};

typedef struct { mod256 P, Qm; mod512 L; u256 w_m; } icc_ctx;   /* w in Montgomery form mod p_icc */

static void mul_256x256(u512 *r, const u256 *a, const u256 *b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        t[i + 4] = (uint64_t)c;
    }
    memcpy(r->l, t, 64);
}
/* w = GENERATOR^((p-1)/(2N)) mod p, Server.hpp:214-216; N a power of two */
static void icc_init(icc_ctx *C, int curve, size_t n) {
    u256 p, q, g, e;
    memcpy(p.l, P_ICC, 32); memcpy(q.l, curve ? Q_SK : Q_BN, 32);
    mod256_init(&C->P, &p); mod256_init(&C->Qm, &q);
    u512 lcm; mul_256x256(&lcm, &p, &q);
    m512_init(&C->L, &lcm);
    u256_from_be(&g, GEN_BE);
    /* e = (p-1) / (2N): p - 1 = 207 * 2^248, shift right by log2(2N) */
    e = p; e.l[0] -= 1;
    int sh = 1; while (((size_t)1 << (sh - 1)) < n) sh++;   /* log2(2N) */
    for (int s = 0; s < sh; s++) { for (int i = 0; i < 3; i++) e.l[i] = (e.l[i] >> 1) | (e.l[i + 1] << 63); e.l[3] >>= 1; }
    u256 gm; mod_to_mont(&C->P, &gm, &g);
    mod_pow(&C->P, &C->w_m, &gm, &e);
}
static void pow_small(const mod256 *M, u256 *r, const u256 *a_m, uint64_t e) { u256 ee = {{e, 0, 0, 0}}; mod_pow(M, r, a_m, &ee); }
static uint64_t rev_bits(uint64_t x, int n) { uint64_t r = 0; for (int i = 0; i < n; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
static void le32_in(u256 *r, const uint8_t *b) { for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int j = 7; j >= 0; j--) w = (w << 8) | b[8 * i + j]; r->l[i] = w; } }
static void le64_out(uint8_t *b, const u512 *a) { for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(a->l[i] >> (8 * j)); }
static void le32_out(uint8_t *b, const u256 *a) { for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(a->l[i] >> (8 * j)); }
static void widen(u512 *r, const u256 *a) { memset(r, 0, sizeof *r); memcpy(r->l, a->l, 32); }

/* A mod m for a 512-bit A and a 256-bit modulus: hi*2^256 + lo */
static void mod512_by_256(const mod256 *M, u256 *r, const u512 *a) {
    u256 hi, lo, t;
    memcpy(lo.l, a->l, 32); memcpy(hi.l, a->l + 4, 32);
    mod256_reduce(M, &hi); mod256_reduce(M, &lo);
    mod_to_mont(M, &t, &hi);          /* hi * 2^256 mod m */
    mod_add(M, r, &t, &lo);
}
/* align_MAC scalar part, Server.hpp:531-541: mod = A % p; c = (mod - A) % q */
static void align_one(const icc_ctx *C, const u512 *A, u256 *mod_out, u256 *c_out) {
    u256 mp, aq, mq;
    mod512_by_256(&C->P, &mp, A);
    mod512_by_256(&C->Qm, &aq, A);
    mq = mp; mod256_reduce(&C->Qm, &mq);
    mod_sub(&C->Qm, c_out, &mq, &aq);
    *mod_out = mp;
}

/* CRebuild_Cached data part for one of X / Y.
 *   rows_in : n * ncols elements, 32-byte LE each (row-major, as U/<i> files, utils.h:592-608)
 *   is_y    : 0 -> X part (copy), 1 -> Y part (element * wt, wt = w^reverse_bits(write_step % n, height-1))
 *   x_out   : n * ncols * 64 bytes LE (values in [0, LCM)), may be NULL
 *   al_out  : n * ncols * 32 bytes LE (values mod p_icc), may be NULL
 *   sc_out  : n * ncols * 32 bytes BE (alignment scalars), may be NULL */
void oracle_icc_crebuild(const uint8_t *rows_in, size_t n, size_t ncols, int curve, int is_y, uint64_t write_step,
                         uint8_t *x_out, uint8_t *al_out, uint8_t *sc_out, int threads) {
    icc_ctx C; icc_init(&C, curve, n);
    int height = 1; while (((size_t)1 << (height - 1)) < n) height++;   /* ceil(log2 n) + 1 */
    u512 *X = (u512 *)malloc(sizeof(u512) * n * ncols);
    u256 wt_m, wt; pow_small(&C.P, &wt_m, &C.w_m, is_y ? rev_bits(write_step % n, height - 1) : 0);
    mod_from_mont(&C.P, &wt, &wt_m);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < n * ncols; i++) {
        u256 x; le32_in(&x, rows_in + 32 * i);
        u512 prod;
        if (is_y) mul_256x256(&prod, &x, &wt); else widen(&prod, &x);     /* Y = X * wt, NOT reduced (Server.hpp:1522) */
        /* enter Montgomery form mod LCM; the value itself may exceed LCM, m512_mul reduces it (a*R2*R^-1 mod LCM) */
        m512_to(&C.L, &X[i], &prod);
    }
    for (int s = 1; s < height; s++) {
        size_t m = (size_t)1 << s, m2 = m >> 1;
        u256 v_m; pow_small(&C.P, &v_m, &C.w_m, n / m2);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
        for (size_t j = 0; j < m2; j++) {
            u256 vi_m, vi; pow_small(&C.P, &vi_m, &v_m, j); mod_from_mont(&C.P, &vi, &vi_m);   /* vi = v^j (Server.hpp:1635) */
            u512 vi_w, vi_L; widen(&vi_w, &vi); m512_to(&C.L, &vi_L, &vi_w);
            for (size_t k = j; k < n; k += m) {
                for (size_t p = 0; p < ncols; p++) {
                    u512 *a = &X[k * ncols + p], *b = &X[(k + m2) * ncols + p], t, u;
                    m512_mul(&C.L, &t, &vi_L, b);      /* t = vi * X[k+m2]  (mod LCM) */
                    u = *a;
                    m512_add(&C.L, a, &u, &t);         /* X[k]    = (u + t) % LCM */
                    m512_sub(&C.L, b, &u, &t);         /* X[k+m2] = (u - t) % LCM */
                }
            }
        }
    }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t i = 0; i < n * ncols; i++) {
        u512 A; m512_from(&C.L, &A, &X[i]);
        if (x_out) le64_out(x_out + 64 * i, &A);
        if (al_out || sc_out) {
            u256 md, c; align_one(&C, &A, &md, &c);
            if (al_out) le32_out(al_out + 32 * i, &md);
            if (sc_out) u256_to_be(sc_out + 32 * i, &c);
        }
    }
    free(X);
}

/* Server::mix data part (Server.hpp:1269-1278): a0, a1: len*ncols elements of 64-byte LE (values < LCM);
 * out: 2*len*ncols elements; v = w^(N/len) with N = n_total */
void oracle_icc_mix(const uint8_t *a0, const uint8_t *a1, size_t len, size_t ncols, size_t n_total, int curve, uint8_t *out) {
    icc_ctx C; icc_init(&C, curve, n_total);
    u256 v_m; pow_small(&C.P, &v_m, &C.w_m, n_total / len);
    u256 vi_m = C.P.r1;
    for (size_t i = 0; i < len; i++) {
        u256 vi; mod_from_mont(&C.P, &vi, &vi_m);
        u512 vi_w, vi_L; widen(&vi_w, &vi); m512_to(&C.L, &vi_L, &vi_w);
        for (size_t p = 0; p < ncols; p++) {
            u512 x0, x1, t, r;
            for (int w = 0; w < 8; w++) { uint64_t a = 0, b = 0; for (int j = 7; j >= 0; j--) { a = (a << 8) | a0[64 * (i * ncols + p) + 8 * w + j]; b = (b << 8) | a1[64 * (i * ncols + p) + 8 * w + j]; } x0.l[w] = a; x1.l[w] = b; }
            m512_to(&C.L, &x0, &x0); m512_to(&C.L, &x1, &x1);
            m512_mul(&C.L, &t, &vi_L, &x1);
            m512_add(&C.L, &r, &x0, &t); m512_from(&C.L, &r, &r); le64_out(out + 64 * (i * ncols + p), &r);
            m512_sub(&C.L, &r, &x0, &t); m512_from(&C.L, &r, &r); le64_out(out + 64 * ((i + len) * ncols + p), &r);
        }
        mod_mul(&C.P, &vi_m, &vi_m, &v_m);
    }
}
