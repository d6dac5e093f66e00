/* ORACLE (test infrastructure only; never linked into the product library).
 *
 * Short-Weierstrass curve y^2 = x^3 + b (a = 0) over a 256-bit prime field: the group law
 * both reference back-ends implement --
 *   BN254 G1  (gnark-crypto v0.6.0 ecc/bn254 g1.go; call sites porla/main.go:59,87,99,136,200,212,220)
 *   secp256k1 (porla/Utils/secp256k1_lib/group_impl.h:274-306 gej_double, :336-387 gej_add_var,
 *              :389-436 gej_add_ge_var)
 * restated with textbook Jacobian formulas.  Parity with the reference is defined on the
 * normalised affine point only, so the choice of projective formulas is free.
 *
 * Also: the multi-scalar multiplication both back-ends expose
 *   (main.go:118-138 -> G1Affine.MultiExp; ecmult_impl.h:814-860 secp256k1_ecmult_multi_var)
 * as (1) a naive sum of double-and-add products and (2) a bucket (Pippenger) method with
 * signed c-bit windows (digits in (-2^(c-1), 2^(c-1)], see the comment at the bucket method below), range-split over threads
 * like Client.hpp:761-787.
 */
#ifndef PORLA_ORACLE_CURVE_A0_H
#define PORLA_ORACLE_CURVE_A0_H
#include "mont256.h"
#include <stdlib.h>

typedef struct { u256 x, y; int inf; } aff_t;      /* coordinates in Montgomery form */
typedef struct { u256 x, y, z; } jac_t;            /* z == 0 <=> infinity */

typedef struct {
    mod256 F;     /* base field */
    u256 b;       /* curve constant, Montgomery form */
} curve_t;

static inline void jac_set_inf(const curve_t *C, jac_t *r) { r->x = C->F.r1; r->y = C->F.r1; memset(&r->z, 0, sizeof(u256)); }
static inline int jac_is_inf(const jac_t *p) { return u256_is_zero(&p->z); }
static inline void jac_from_aff(const curve_t *C, jac_t *r, const aff_t *a) {
    if (a->inf) { jac_set_inf(C, r); return; }
    r->x = a->x; r->y = a->y; r->z = C->F.r1;
}
static inline void jac_neg(const curve_t *C, jac_t *r, const jac_t *p) { *r = *p; mod_neg(&C->F, &r->y, &p->y); }

static inline void jac_double(const curve_t *C, jac_t *r, const jac_t *p) {
    const mod256 *F = &C->F;
    if (jac_is_inf(p) || u256_is_zero(&p->y)) { jac_set_inf(C, r); return; }
    u256 A, B, Cc, D, E, Fq, t, X3, Y3, Z3;
    mod_sqr(F, &A, &p->x);
    mod_sqr(F, &B, &p->y);
    mod_sqr(F, &Cc, &B);
    mod_add(F, &t, &p->x, &B); mod_sqr(F, &t, &t); mod_sub(F, &t, &t, &A); mod_sub(F, &t, &t, &Cc);
    mod_add(F, &D, &t, &t);
    mod_add(F, &E, &A, &A); mod_add(F, &E, &E, &A);
    mod_sqr(F, &Fq, &E);
    mod_sub(F, &X3, &Fq, &D); mod_sub(F, &X3, &X3, &D);
    mod_sub(F, &t, &D, &X3); mod_mul(F, &Y3, &E, &t);
    mod_add(F, &t, &Cc, &Cc); mod_add(F, &t, &t, &t); mod_add(F, &t, &t, &t);
    mod_sub(F, &Y3, &Y3, &t);
    mod_mul(F, &Z3, &p->y, &p->z); mod_add(F, &Z3, &Z3, &Z3);
    r->x = X3; r->y = Y3; r->z = Z3;
}

/* r = p + q, both Jacobian, all special cases handled */
static inline void jac_add(const curve_t *C, jac_t *r, const jac_t *p, const jac_t *q) {
    const mod256 *F = &C->F;
    if (jac_is_inf(p)) { *r = *q; return; }
    if (jac_is_inf(q)) { *r = *p; return; }
    u256 Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, t, HH, HHH, V, X3, Y3, Z3;
    mod_sqr(F, &Z1Z1, &p->z); mod_sqr(F, &Z2Z2, &q->z);
    mod_mul(F, &U1, &p->x, &Z2Z2); mod_mul(F, &U2, &q->x, &Z1Z1);
    mod_mul(F, &t, &q->z, &Z2Z2); mod_mul(F, &S1, &p->y, &t);
    mod_mul(F, &t, &p->z, &Z1Z1); mod_mul(F, &S2, &q->y, &t);
    mod_sub(F, &H, &U2, &U1); mod_sub(F, &Rr, &S2, &S1);
    if (u256_is_zero(&H)) {
        if (u256_is_zero(&Rr)) { jac_double(C, r, p); return; }
        jac_set_inf(C, r); return;
    }
    mod_sqr(F, &HH, &H); mod_mul(F, &HHH, &HH, &H); mod_mul(F, &V, &U1, &HH);
    mod_sqr(F, &X3, &Rr); mod_sub(F, &X3, &X3, &HHH); mod_sub(F, &X3, &X3, &V); mod_sub(F, &X3, &X3, &V);
    mod_sub(F, &t, &V, &X3); mod_mul(F, &Y3, &Rr, &t); mod_mul(F, &t, &S1, &HHH); mod_sub(F, &Y3, &Y3, &t);
    mod_mul(F, &Z3, &p->z, &q->z); mod_mul(F, &Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

/* r = p + a (a affine) */
static inline void jac_add_aff(const curve_t *C, jac_t *r, const jac_t *p, const aff_t *a) {
    const mod256 *F = &C->F;
    if (a->inf) { *r = *p; return; }
    if (jac_is_inf(p)) { jac_from_aff(C, r, a); return; }
    u256 Z1Z1, U2, S2, H, Rr, t, HH, HHH, V, X3, Y3, Z3;
    mod_sqr(F, &Z1Z1, &p->z);
    mod_mul(F, &U2, &a->x, &Z1Z1);
    mod_mul(F, &t, &p->z, &Z1Z1); mod_mul(F, &S2, &a->y, &t);
    mod_sub(F, &H, &U2, &p->x); mod_sub(F, &Rr, &S2, &p->y);
    if (u256_is_zero(&H)) {
        if (u256_is_zero(&Rr)) { jac_double(C, r, p); return; }
        jac_set_inf(C, r); return;
    }
    mod_sqr(F, &HH, &H); mod_mul(F, &HHH, &HH, &H); mod_mul(F, &V, &p->x, &HH);
    mod_sqr(F, &X3, &Rr); mod_sub(F, &X3, &X3, &HHH); mod_sub(F, &X3, &X3, &V); mod_sub(F, &X3, &X3, &V);
    mod_sub(F, &t, &V, &X3); mod_mul(F, &Y3, &Rr, &t); mod_mul(F, &t, &p->y, &HHH); mod_sub(F, &Y3, &Y3, &t);
    mod_mul(F, &Z3, &p->z, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static inline void jac_to_aff(const curve_t *C, aff_t *r, const jac_t *p) {
    const mod256 *F = &C->F;
    if (jac_is_inf(p)) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }
    u256 zi, zi2, zi3;
    mod_inv(F, &zi, &p->z); mod_sqr(F, &zi2, &zi); mod_mul(F, &zi3, &zi2, &zi);
    mod_mul(F, &r->x, &p->x, &zi2); mod_mul(F, &r->y, &p->y, &zi3); r->inf = 0;
}

/* y^2 == x^3 + b ? */
static inline int aff_on_curve(const curve_t *C, const aff_t *a) {
    if (a->inf) return 1;
    u256 l, rr;
    mod_sqr(&C->F, &l, &a->y);
    mod_sqr(&C->F, &rr, &a->x); mod_mul(&C->F, &rr, &rr, &a->x); mod_add(&C->F, &rr, &rr, &C->b);
    return u256_eq(&l, &rr);
}

/* r = k * a, k a plain 256-bit integer (left-to-right double-and-add) */
static inline void jac_mul_aff(const curve_t *C, jac_t *r, const aff_t *a, const u256 *k) {
    jac_t acc; jac_set_inf(C, &acc);
    for (int i = 255; i >= 0; i--) {
        jac_double(C, &acc, &acc);
        if (u256_bit(k, i)) jac_add_aff(C, &acc, &acc, a);
    }
    *r = acc;
}

/* naive MSM: sum k_i * P_i */
static inline void msm_naive(const curve_t *C, jac_t *r, const u256 *k, const aff_t *pts, size_t n) {
    jac_t acc, t; jac_set_inf(C, &acc);
    for (size_t i = 0; i < n; i++) { jac_mul_aff(C, &t, &pts[i], &k[i]); jac_add(C, &acc, &acc, &t); }
    *r = acc;
}

static inline int msm_window_bits(size_t n) {
    int c = 1; while (((size_t)1 << (c + 4)) < n * 3 && c < 16) c++;  /* ~ log2(n) - 2, capped */
    if (c < 2) c = 2;
    return c;
}

/* bucket MSM over one contiguous range: SIGNED c-bit windows over `nbits` scalar bits (digits in (-2^(c-1), 2^(c-1)]: a digit
 * above 2^(c-1) becomes digit - 2^c with a carry into the next window, a negative digit subtracts the point), 2^(c-1) buckets
 * per window -- the window is one bit wider than an unsigned one at the same bucket count, as in gnark's and libsecp256k1's
 * own bucket methods (signed digits / wNAF, ecmult_impl.h:492-567).  The group element computed is the same. */
static inline void msm_pippenger_range(const curve_t *C, jac_t *r, const u256 *k, const aff_t *pts,
                                       size_t n, int nbits) {
    int c = msm_window_bits(n) + 1;
    size_t nb = (size_t)1 << (c - 1);
    int nwin = (nbits + 1 + c - 1) / c;
    jac_t *bucket = (jac_t *)malloc(sizeof(jac_t) * nb);
    int32_t *dig = (int32_t *)malloc(sizeof(int32_t) * (size_t)nwin * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        uint32_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            uint32_t raw = u256_bits(&k[i], w * c, c) + carry;            /* bits above 255 read as zero */
            if (raw > nb) { dig[(size_t)w * n + i] = (int32_t)raw - (int32_t)((uint32_t)1 << c); carry = 1; }
            else { dig[(size_t)w * n + i] = (int32_t)raw; carry = 0; }
        }
    }
    jac_t total; jac_set_inf(C, &total);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int d = 0; d < c; d++) jac_double(C, &total, &total);
        for (size_t b = 0; b < nb; b++) jac_set_inf(C, &bucket[b]);
        for (size_t i = 0; i < n; i++) {
            int32_t d = dig[(size_t)w * n + i];
            if (d > 0) jac_add_aff(C, &bucket[d - 1], &bucket[d - 1], &pts[i]);
            else if (d < 0 && !pts[i].inf) {
                aff_t np = pts[i];
                mod_neg(&C->F, &np.y, &pts[i].y);
                jac_add_aff(C, &bucket[-d - 1], &bucket[-d - 1], &np);
            }
        }
        jac_t run, sum; jac_set_inf(C, &run); jac_set_inf(C, &sum);
        for (size_t b = nb; b-- > 0;) { jac_add(C, &run, &run, &bucket[b]); jac_add(C, &sum, &sum, &run); }
        jac_add(C, &total, &total, &sum);
    }
    free(bucket);
    free(dig);
    *r = total;
}

/* threads: input range-split, partial sums added (the reference's own strategy, Client.hpp:761-787) */
static inline void msm_pippenger(const curve_t *C, jac_t *r, const u256 *k, const aff_t *pts, size_t n,
                                 int nbits, int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n / 64 + 1) threads = (int)(n / 64 + 1);
    jac_t *part = (jac_t *)malloc(sizeof(jac_t) * threads);
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int t = 0; t < threads; t++) {
        size_t lo = n * (size_t)t / threads, hi = n * (size_t)(t + 1) / threads;
        msm_pippenger_range(C, &part[t], k + lo, pts + lo, hi - lo, nbits);
    }
    jac_t acc; jac_set_inf(C, &acc);
    for (int t = 0; t < threads; t++) jac_add(C, &acc, &acc, &part[t]);
    free(part);
    *r = acc;
}

/* batch normalisation (Montgomery's trick) of n Jacobian points into affine */
static inline void jac_batch_to_aff(const curve_t *C, aff_t *out, const jac_t *in, size_t n) {
    const mod256 *F = &C->F;
    u256 *pre = (u256 *)malloc(sizeof(u256) * (n + 1));
    u256 acc = F->r1;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!jac_is_inf(&in[i])) mod_mul(F, &acc, &acc, &in[i].z); }
    u256 inv; mod_inv(F, &inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (jac_is_inf(&in[i])) { memset(&out[i], 0, sizeof(aff_t)); out[i].inf = 1; continue; }
        u256 zi, zi2, zi3;
        mod_mul(F, &zi, &inv, &pre[i]); mod_mul(F, &inv, &inv, &in[i].z);
        mod_sqr(F, &zi2, &zi); mod_mul(F, &zi3, &zi2, &zi);
        mod_mul(F, &out[i].x, &in[i].x, &zi2); mod_mul(F, &out[i].y, &in[i].y, &zi3); out[i].inf = 0;
    }
    free(pre);
}
#endif
