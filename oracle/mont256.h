/* ORACLE (test infrastructure only; never linked into the product library).
 *
 * Generic 256-bit modular arithmetic, 4 x 64-bit limbs, Montgomery form with R = 2^256,
 * written in plain C with unsigned __int128.  It restates, for any odd modulus < 2^256,
 * what the reference obtains from its third-party providers:
 *   - gnark-crypto v0.6.0 fp/fr (4 x 64 Montgomery)      -- call sites porla/main.go:34-214
 *   - libsecp256k1 field_5x52 / scalar_4x64               -- porla/Utils/secp256k1_lib/field_5x52_impl.h:432,
 *                                                            scalar_4x64_impl.h:733
 *   - NTL ZZ_p arithmetic on p_icc                        -- porla/Utils/utils.h:31-43
 * Internal representation is irrelevant to parity: every oracle entry point takes and
 * returns canonical byte strings.
 */
#ifndef PORLA_ORACLE_MONT256_H
#define PORLA_ORACLE_MONT256_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } u256;

typedef struct {
    u256 m;        /* modulus */
    uint64_t inv;  /* -m^-1 mod 2^64 */
    u256 r1;       /* R mod m   (Montgomery one) */
    u256 r2;       /* R^2 mod m */
} mod256;

static inline int u256_is_zero(const u256 *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int u256_eq(const u256 *a, const u256 *b) { return memcmp(a, b, sizeof(u256)) == 0; }
static inline int u256_geq(const u256 *a, const u256 *b) {
    for (int i = 3; i >= 0; i--) { if (a->l[i] != b->l[i]) return a->l[i] > b->l[i]; }
    return 1;
}
static inline uint64_t u256_add(u256 *r, const u256 *a, const u256 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t u256_sub(u256 *r, const u256 *a, const u256 *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline void u256_from_be(u256 *r, const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w = (w << 8) | b[(3 - i) * 8 + j];
        r->l[i] = w;
    }
}
static inline void u256_to_be(uint8_t b[32], const u256 *a) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(a->l[i] >> (56 - 8 * j));
}
static inline int u256_bit(const u256 *a, int i) { return (a->l[i >> 6] >> (i & 63)) & 1; }
/* bits [lo, lo+n) as an unsigned integer, n <= 32 */
static inline uint32_t u256_bits(const u256 *a, int lo, int n) {
    if (lo >= 256) return 0;
    int w = lo >> 6, s = lo & 63;
    uint64_t v = a->l[w] >> s;
    if (s + n > 64 && w < 3) v |= a->l[w + 1] << (64 - s);
    return (uint32_t)(v & ((n >= 32) ? 0xffffffffu : ((1u << n) - 1)));
}

/* plain reduction of a < 2^256 modulo m (repeated subtraction is enough for the moduli used here:
 * 2^256 / m < 6 for BN254 p and r, < 2 for secp256k1 p, n and for p_icc) */
static inline void mod256_reduce(const mod256 *M, u256 *a) {
    while (u256_geq(a, &M->m)) u256_sub(a, a, &M->m);
}
static inline void mod_add(const mod256 *M, u256 *r, const u256 *a, const u256 *b) {
    uint64_t c = u256_add(r, a, b);
    if (c || u256_geq(r, &M->m)) u256_sub(r, r, &M->m);
}
static inline void mod_sub(const mod256 *M, u256 *r, const u256 *a, const u256 *b) {
    if (u256_sub(r, a, b)) u256_add(r, r, &M->m);
}
static inline void mod_neg(const mod256 *M, u256 *r, const u256 *a) {
    if (u256_is_zero(a)) { *r = *a; return; }
    u256_sub(r, &M->m, a);
}
/* Montgomery product a*b*R^-1 mod m (CIOS) */
static inline void mod_mul(const mod256 *M, u256 *r, const u256 *a, const u256 *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * M->inv;
        c = (u128)q * M->m.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)q * M->m.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    u256 x = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || u256_geq(&x, &M->m)) u256_sub(&x, &x, &M->m);
    *r = x;
}
static inline void mod_sqr(const mod256 *M, u256 *r, const u256 *a) { mod_mul(M, r, a, a); }
static inline void mod_to_mont(const mod256 *M, u256 *r, const u256 *a) { mod_mul(M, r, a, &M->r2); }
static inline void mod_from_mont(const mod256 *M, u256 *r, const u256 *a) {
    u256 one = {{1, 0, 0, 0}};
    mod_mul(M, r, a, &one);
}
/* r = a^e (a, r Montgomery form; e a plain integer) */
static inline void mod_pow(const mod256 *M, u256 *r, const u256 *a, const u256 *e) {
    u256 acc = M->r1, base = *a;
    for (int i = 255; i >= 0; i--) {
        mod_sqr(M, &acc, &acc);
        if (u256_bit(e, i)) mod_mul(M, &acc, &acc, &base);
    }
    *r = acc;
}
static inline void mod_inv(const mod256 *M, u256 *r, const u256 *a) { /* prime modulus: a^(m-2) */
    u256 e = M->m, two = {{2, 0, 0, 0}};
    u256_sub(&e, &e, &two);
    mod_pow(M, r, a, &e);
}
/* derive inv, r1, r2 from the modulus (so no magic constants have to be trusted) */
static inline void mod256_init(mod256 *M, const u256 *m) {
    M->m = *m;
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - m->l[0] * x; /* Newton: m^-1 mod 2^64 */
    M->inv = (uint64_t)0 - x;
    /* r1 = 2^256 mod m by 256 modular doublings of 1; r2 by 256 more */
    u256 v = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        uint64_t c = u256_add(&v, &v, &v);
        if (c || u256_geq(&v, m)) u256_sub(&v, &v, m);
        if (i == 255) M->r1 = v;
    }
    M->r2 = v;
}
#endif
