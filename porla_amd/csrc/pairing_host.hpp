// Host-only BN254 helpers needed to keep the full 14-symbol surface of libmultiexp.so:
//   * G1 / G2 compressed (de)serialisation of the SRS wire blob  (kzg.SRS.WriteTo / ReadFrom, main.go:48,67)
//   * G2 scalar multiplication for SRS.G2[1] = tau * G2gen        (kzg.NewSRS, main.go:46)
//   * a pairing-product check for kzg.Verify                       (main.go:187)
// None of this is on the data-parallel hot path (one call per audit, Client.hpp:1635-1663): Fp12 = Fp2[w]/(w^6 - xi),
// xi = 9 + u, the plain ate pairing f_{t-1,Q}(P) with inversion-free (projective) line steps, and the final exponentiation
// split as usual: easy part f^((p^6-1)(p^2+1)) by conjugation, one inversion and a Frobenius, hard part f^((p^4-p^2+1)/r)
// by the BN addition chain over f^x, f^(x^2), f^(x^3) and Frobenius maps (exponent identity checked in
// tests/test_constants.py).  The literal exponentiation by (p^12-1)/r is kept as the slow reference (pairing self-check).
// kzg.Verify only tests e(A,G2)*e(-H,Q) == 1, and every non-degenerate bilinear pairing on G1 x G2 agrees on that predicate.
#pragma once
#include "host_curve.hpp"
#include <vector>

namespace porla {

using FpE = Fe<Bn254Fp>;

inline FpE fp_small(uint32_t v) { FpE a = fe_zero<Bn254Fp>(); a.v[0] = v; return fe_to_mont<Bn254Fp>(a); }

// regular value > (p-1)/2 ?   (fp.Element.LexicographicallyLargest)
inline bool fp_lex_largest(const FpE& a) {
    uint32_t v[8];
    h_fe_to_plain<Bn254Fp>(v, a);
    // half = (p-1)/2
    uint32_t half[8];
    for (int i = 0; i < 8; i++) half[i] = (Bn254Fp::P[i] >> 1) | (i < 7 ? (Bn254Fp::P[i + 1] << 31) : 0);
    for (int i = 7; i >= 0; i--) { if (v[i] != half[i]) return v[i] > half[i]; }
    return false;
}
// sqrt in Fp (p = 3 mod 4): a^((p+1)/4); returns false if a is not a square
inline bool fp_sqrt(FpE* r, const FpE& a) {
    uint32_t e[8];
    uint64_t c = 1;
    for (int i = 0; i < 8; i++) { uint64_t s = (uint64_t)Bn254Fp::P[i] + c; e[i] = (uint32_t)s; c = s >> 32; }
    for (int i = 0; i < 8; i++) e[i] = (e[i] >> 2) | (i < 7 ? (e[i + 1] << 30) : 0);
    FpE s = h_fe_pow<Bn254Fp>(a, e);
    *r = s;
    return fe_eq<Bn254Fp>(fe_sqr<Bn254Fp>(s), a);
}

// ---------------------------------------------------------------- G1 compressed form (gnark marshal.go)
inline void g1_compress(uint8_t out[32], const Affine<Bn254Fp>& a) {
    if (aff_is_inf<Bn254Fp>(a)) { memset(out, 0, 32); out[0] = 0x40; return; }
    h_fe_to_be<Bn254Fp>(out, a.x);
    out[0] |= fp_lex_largest(a.y) ? 0xC0 : 0x80;
}
inline bool g1_decompress(const uint8_t in[32], Affine<Bn254Fp>* a) {
    uint8_t flags = in[0] & 0xC0;
    if (flags == 0x40) { a->x = fe_zero<Bn254Fp>(); a->y = fe_zero<Bn254Fp>(); return true; }
    uint8_t t[32];
    memcpy(t, in, 32);
    t[0] &= 0x3F;
    a->x = h_fe_from_be<Bn254Fp>(t);
    FpE rhs = fe_add<Bn254Fp>(fe_mul<Bn254Fp>(fe_sqr<Bn254Fp>(a->x), a->x), fp_small(3));
    FpE y;
    if (!fp_sqrt(&y, rhs)) return false;
    if (fp_lex_largest(y) != (flags == 0xC0)) y = fe_neg<Bn254Fp>(y);
    a->y = y;
    return true;
}

// ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2 + 1)
struct Fp2 { FpE a0, a1; };
inline Fp2 f2_zero() { return Fp2{fe_zero<Bn254Fp>(), fe_zero<Bn254Fp>()}; }
inline Fp2 f2_one() { return Fp2{fe_one<Bn254Fp>(), fe_zero<Bn254Fp>()}; }
inline bool f2_is_zero(const Fp2& a) { return fe_is_zero<Bn254Fp>(a.a0) && fe_is_zero<Bn254Fp>(a.a1); }
inline bool f2_eq(const Fp2& a, const Fp2& b) { return fe_eq<Bn254Fp>(a.a0, b.a0) && fe_eq<Bn254Fp>(a.a1, b.a1); }
inline Fp2 f2_add(const Fp2& a, const Fp2& b) { return Fp2{fe_add<Bn254Fp>(a.a0, b.a0), fe_add<Bn254Fp>(a.a1, b.a1)}; }
inline Fp2 f2_sub(const Fp2& a, const Fp2& b) { return Fp2{fe_sub<Bn254Fp>(a.a0, b.a0), fe_sub<Bn254Fp>(a.a1, b.a1)}; }
inline Fp2 f2_neg(const Fp2& a) { return Fp2{fe_neg<Bn254Fp>(a.a0), fe_neg<Bn254Fp>(a.a1)}; }
inline Fp2 f2_mul(const Fp2& a, const Fp2& b) {
    FpE t0 = fe_mul<Bn254Fp>(a.a0, b.a0), t1 = fe_mul<Bn254Fp>(a.a1, b.a1);
    FpE s = fe_mul<Bn254Fp>(fe_add<Bn254Fp>(a.a0, a.a1), fe_add<Bn254Fp>(b.a0, b.a1));
    return Fp2{fe_sub<Bn254Fp>(t0, t1), fe_sub<Bn254Fp>(fe_sub<Bn254Fp>(s, t0), t1)};
}
inline Fp2 f2_sqr(const Fp2& a) {   // (a0 + a1)(a0 - a1) + 2 a0 a1 u
    FpE t = fe_mul<Bn254Fp>(a.a0, a.a1);
    return Fp2{fe_mul<Bn254Fp>(fe_add<Bn254Fp>(a.a0, a.a1), fe_sub<Bn254Fp>(a.a0, a.a1)), fe_dbl<Bn254Fp>(t)};
}
inline Fp2 f2_mul_fp(const Fp2& a, const FpE& k) { return Fp2{fe_mul<Bn254Fp>(a.a0, k), fe_mul<Bn254Fp>(a.a1, k)}; }
inline Fp2 f2_inv(const Fp2& a) {
    FpE n = fe_add<Bn254Fp>(fe_sqr<Bn254Fp>(a.a0), fe_sqr<Bn254Fp>(a.a1));
    FpE i = h_fe_inv<Bn254Fp>(n);
    return Fp2{fe_mul<Bn254Fp>(a.a0, i), fe_neg<Bn254Fp>(fe_mul<Bn254Fp>(a.a1, i))};
}
inline Fp2 f2_xi() { return Fp2{fp_small(9), fp_small(1)}; }  // 9 + u
inline Fp2 f2_conj(const Fp2& a) { return Fp2{a.a0, fe_neg<Bn254Fp>(a.a1)}; }
inline Fp2 f2_mul_xi(const Fp2& a) {  // (a0 + a1 u)(9 + u) = (9 a0 - a1) + (a0 + 9 a1) u
    FpE a0_8 = fe_dbl<Bn254Fp>(fe_dbl<Bn254Fp>(fe_dbl<Bn254Fp>(a.a0))), a1_8 = fe_dbl<Bn254Fp>(fe_dbl<Bn254Fp>(fe_dbl<Bn254Fp>(a.a1)));
    return Fp2{fe_sub<Bn254Fp>(fe_add<Bn254Fp>(a0_8, a.a0), a.a1), fe_add<Bn254Fp>(fe_add<Bn254Fp>(a1_8, a.a1), a.a0)};
}
inline Fp2 f2_pow(const Fp2& a, const uint32_t e[8]) {
    Fp2 acc = Fp2{fe_one<Bn254Fp>(), fe_zero<Bn254Fp>()};
    for (int i = 255; i >= 0; i--) {
        acc = f2_mul(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = f2_mul(acc, a);
    }
    return acc;
}
inline bool f2_lex_largest(const Fp2& a) {  // E2.LexicographicallyLargest
    if (fe_is_zero<Bn254Fp>(a.a1)) return fp_lex_largest(a.a0);
    return fp_lex_largest(a.a1);
}
inline bool f2_sqrt(Fp2* r, const Fp2& a) {
    if (f2_is_zero(a)) { *r = a; return true; }
    FpE half = h_fe_inv<Bn254Fp>(fp_small(2));
    Fp2 cand;
    if (fe_is_zero<Bn254Fp>(a.a1)) {
        FpE s;
        if (fp_sqrt(&s, a.a0)) cand = Fp2{s, fe_zero<Bn254Fp>()};
        else { if (!fp_sqrt(&s, fe_neg<Bn254Fp>(a.a0))) return false; cand = Fp2{fe_zero<Bn254Fp>(), s}; }
    } else {
        FpE norm = fe_add<Bn254Fp>(fe_sqr<Bn254Fp>(a.a0), fe_sqr<Bn254Fp>(a.a1));
        FpE alpha;
        if (!fp_sqrt(&alpha, norm)) return false;
        FpE delta = fe_mul<Bn254Fp>(fe_add<Bn254Fp>(a.a0, alpha), half);
        FpE x0;
        if (!fp_sqrt(&x0, delta)) {
            delta = fe_mul<Bn254Fp>(fe_sub<Bn254Fp>(a.a0, alpha), half);
            if (!fp_sqrt(&x0, delta)) return false;
        }
        FpE x1 = fe_mul<Bn254Fp>(fe_mul<Bn254Fp>(a.a1, half), h_fe_inv<Bn254Fp>(x0));
        cand = Fp2{x0, x1};
    }
    *r = cand;
    return f2_eq(f2_sqr(cand), a);
}

// ---------------------------------------------------------------- G2: E'(Fp2): y^2 = x^3 + 3/xi, affine
struct G2Affine { Fp2 x, y; bool inf; };
inline Fp2 g2_b() { return f2_mul(Fp2{fp_small(3), fe_zero<Bn254Fp>()}, f2_inv(f2_xi())); }
inline FpE fp_from_hex_be(const char* hex) {  // 64 hex digits
    uint8_t b[32];
    for (int i = 0; i < 32; i++) {
        auto nib = [](char c) -> int { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; };
        b[i] = (uint8_t)((nib(hex[2 * i]) << 4) | nib(hex[2 * i + 1]));
    }
    return h_fe_from_be<Bn254Fp>(b);
}
inline G2Affine g2_generator() {  // the standard alt_bn128 G2 generator (EIP-197; gnark g2Gen)
    G2Affine g;
    g.x.a0 = fp_from_hex_be("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed");
    g.x.a1 = fp_from_hex_be("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2");
    g.y.a0 = fp_from_hex_be("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa");
    g.y.a1 = fp_from_hex_be("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b");
    g.inf = false;
    return g;
}
inline G2Affine g2_neg(const G2Affine& p) { G2Affine r = p; if (!p.inf) r.y = f2_neg(p.y); return r; }
inline G2Affine g2_double(const G2Affine& p) {
    if (p.inf || f2_is_zero(p.y)) return G2Affine{f2_zero(), f2_zero(), true};
    Fp2 xx = f2_sqr(p.x);
    Fp2 lam = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(p.y, p.y)));
    G2Affine r;
    r.x = f2_sub(f2_sub(f2_sqr(lam), p.x), p.x);
    r.y = f2_sub(f2_mul(lam, f2_sub(p.x, r.x)), p.y);
    r.inf = false;
    return r;
}
inline G2Affine g2_add(const G2Affine& p, const G2Affine& q) {
    if (p.inf) return q;
    if (q.inf) return p;
    if (f2_eq(p.x, q.x)) {
        if (f2_eq(p.y, q.y)) return g2_double(p);
        return G2Affine{f2_zero(), f2_zero(), true};
    }
    Fp2 lam = f2_mul(f2_sub(q.y, p.y), f2_inv(f2_sub(q.x, p.x)));
    G2Affine r;
    r.x = f2_sub(f2_sub(f2_sqr(lam), p.x), q.x);
    r.y = f2_sub(f2_mul(lam, f2_sub(p.x, r.x)), p.y);
    r.inf = false;
    return r;
}
inline G2Affine g2_scalar_mul(const G2Affine& p, const uint32_t k[8]) {
    G2Affine acc{f2_zero(), f2_zero(), true};
    for (int i = 255; i >= 0; i--) {
        acc = g2_double(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = g2_add(acc, p);
    }
    return acc;
}
// G2Affine.Bytes(): X.A1 || X.A0 big-endian, flags in the top two bits of byte 0
inline void g2_compress(uint8_t out[64], const G2Affine& p) {
    if (p.inf) { memset(out, 0, 64); out[0] = 0x40; return; }
    h_fe_to_be<Bn254Fp>(out, p.x.a1);
    h_fe_to_be<Bn254Fp>(out + 32, p.x.a0);
    out[0] |= f2_lex_largest(p.y) ? 0xC0 : 0x80;
}
inline bool g2_decompress(const uint8_t in[64], G2Affine* p) {
    uint8_t flags = in[0] & 0xC0;
    if (flags == 0x40) { *p = G2Affine{f2_zero(), f2_zero(), true}; return true; }
    uint8_t t[32];
    memcpy(t, in, 32);
    t[0] &= 0x3F;
    p->x.a1 = h_fe_from_be<Bn254Fp>(t);
    p->x.a0 = h_fe_from_be<Bn254Fp>(in + 32);
    Fp2 rhs = f2_add(f2_mul(f2_sqr(p->x), p->x), g2_b());
    Fp2 y;
    if (!f2_sqrt(&y, rhs)) return false;
    if (f2_lex_largest(y) != (flags == 0xC0)) y = f2_neg(y);
    p->y = y;
    p->inf = false;
    return true;
}

// ---------------------------------------------------------------- Fp12 = Fp2[w]/(w^6 - xi)
struct Fp12 { Fp2 c[6]; };
inline Fp12 f12_one() { Fp12 r; for (int i = 0; i < 6; i++) r.c[i] = f2_zero(); r.c[0] = f2_one(); return r; }
// Fp6 = Fp2[v]/(v^3 - xi), v = w^2; an Fp12 element is A + w B with A = (c0, c2, c4), B = (c1, c3, c5)
struct Fp6 { Fp2 a0, a1, a2; };
inline Fp6 f6_add(const Fp6& a, const Fp6& b) { return Fp6{f2_add(a.a0, b.a0), f2_add(a.a1, b.a1), f2_add(a.a2, b.a2)}; }
inline Fp6 f6_sub(const Fp6& a, const Fp6& b) { return Fp6{f2_sub(a.a0, b.a0), f2_sub(a.a1, b.a1), f2_sub(a.a2, b.a2)}; }
inline Fp6 f6_mul_v(const Fp6& a) { return Fp6{f2_mul_xi(a.a2), a.a0, a.a1}; }
inline Fp6 f6_mul(const Fp6& a, const Fp6& b) {   // Karatsuba: 6 products in Fp2
    Fp2 t0 = f2_mul(a.a0, b.a0), t1 = f2_mul(a.a1, b.a1), t2 = f2_mul(a.a2, b.a2);
    Fp2 c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a.a1, a.a2), f2_add(b.a1, b.a2)), t1), t2)));
    Fp2 c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.a0, a.a1), f2_add(b.a0, b.a1)), t0), t1), f2_mul_xi(t2));
    Fp2 c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.a0, a.a2), f2_add(b.a0, b.a2)), t0), t2), t1);
    return Fp6{c0, c1, c2};
}
inline void f12_split(const Fp12& a, Fp6* A, Fp6* B) { *A = Fp6{a.c[0], a.c[2], a.c[4]}; *B = Fp6{a.c[1], a.c[3], a.c[5]}; }
inline Fp12 f12_join(const Fp6& A, const Fp6& B) {
    Fp12 r;
    r.c[0] = A.a0; r.c[2] = A.a1; r.c[4] = A.a2; r.c[1] = B.a0; r.c[3] = B.a1; r.c[5] = B.a2;
    return r;
}
// (A0 + w A1)(B0 + w B1) = (A0 B0 + v A1 B1) + w ((A0 + A1)(B0 + B1) - A0 B0 - A1 B1): 3 products in Fp6 = 18 in Fp2 (the
// schoolbook form over the six Fp2 coefficients took 36, plus five products by xi that are additions in f2_mul_xi)
inline Fp12 f12_mul(const Fp12& a, const Fp12& b) {
    Fp6 A0, A1, B0, B1;
    f12_split(a, &A0, &A1); f12_split(b, &B0, &B1);
    const Fp6 t0 = f6_mul(A0, B0), t1 = f6_mul(A1, B1), t2 = f6_mul(f6_add(A0, A1), f6_add(B0, B1));
    return f12_join(f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(t2, t0), t1));
}
// (A0 + w A1)^2 = ((A0 + A1)(A0 + v A1) - t - v t) + w 2t, t = A0 A1: 2 products in Fp6
inline Fp12 f12_sqr(const Fp12& a) {
    Fp6 A0, A1;
    f12_split(a, &A0, &A1);
    const Fp6 t = f6_mul(A0, A1);
    const Fp6 r0 = f6_sub(f6_sub(f6_mul(f6_add(A0, A1), f6_add(A0, f6_mul_v(A1))), t), f6_mul_v(t));
    return f12_join(r0, f6_add(t, t));
}
// a times a SPARSE element (a line value: three of its six coefficients are zero): schoolbook over the non-zero coefficients
inline Fp12 f12_mul_sparse(const Fp12& a, const Fp12& b) {
    Fp2 t[11];
    for (int i = 0; i < 11; i++) t[i] = f2_zero();
    for (int j = 0; j < 6; j++) {
        if (f2_is_zero(b.c[j])) continue;
        for (int i = 0; i < 6; i++) t[i + j] = f2_add(t[i + j], f2_mul(a.c[i], b.c[j]));
    }
    Fp12 r;
    for (int i = 0; i < 6; i++) r.c[i] = (i + 6 < 11) ? f2_add(t[i], f2_mul_xi(t[i + 6])) : t[i];
    return r;
}
inline bool f12_is_one(const Fp12& a) {
    if (!f2_eq(a.c[0], f2_one())) return false;
    for (int i = 1; i < 6; i++) if (!f2_is_zero(a.c[i])) return false;
    return true;
}

// conjugation = the p^6 Frobenius: w -> -w
inline Fp12 f12_conj(const Fp12& a) {
    Fp12 r = a;
    r.c[1] = f2_neg(a.c[1]); r.c[3] = f2_neg(a.c[3]); r.c[5] = f2_neg(a.c[5]);
    return r;
}
inline Fp6 f6_inv(const Fp6& a) {
    Fp2 t0 = f2_sub(f2_sqr(a.a0), f2_mul_xi(f2_mul(a.a1, a.a2)));
    Fp2 t1 = f2_sub(f2_mul_xi(f2_sqr(a.a2)), f2_mul(a.a0, a.a1));
    Fp2 t2 = f2_sub(f2_sqr(a.a1), f2_mul(a.a0, a.a2));
    Fp2 d = f2_add(f2_mul(a.a0, t0), f2_mul_xi(f2_add(f2_mul(a.a2, t1), f2_mul(a.a1, t2))));
    Fp2 di = f2_inv(d);
    return Fp6{f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)};
}
// a = A + w B with A = (c0, c2, c4), B = (c1, c3, c5) in Fp6, w^2 = v:  a^-1 = (A - w B) / (A^2 - v B^2)
inline Fp12 f12_inv(const Fp12& a) {
    Fp6 A{a.c[0], a.c[2], a.c[4]}, B{a.c[1], a.c[3], a.c[5]};
    Fp6 d = f6_inv(f6_sub(f6_mul(A, A), f6_mul_v(f6_mul(B, B))));
    Fp6 ra = f6_mul(A, d), rb = f6_mul(B, d);
    Fp12 r;
    r.c[0] = ra.a0; r.c[2] = ra.a1; r.c[4] = ra.a2;
    r.c[1] = f2_neg(rb.a0); r.c[3] = f2_neg(rb.a1); r.c[5] = f2_neg(rb.a2);
    return r;
}
// p-power Frobenius: (sum c_i w^i)^p = sum conj(c_i) * gamma^i * w^i with gamma = xi^((p-1)/6)
inline const Fp2* f12_frob_gammas() {
    static Fp2 g[6];
    static bool ready = false;
    if (!ready) {
        uint32_t e[8];   // (p - 1) / 6
        uint64_t rem = 0;
        uint32_t pm1[8];
        for (int i = 0; i < 8; i++) pm1[i] = Bn254Fp::P[i];
        pm1[0] -= 1;
        for (int i = 7; i >= 0; i--) { uint64_t cur = (rem << 32) | pm1[i]; e[i] = (uint32_t)(cur / 6); rem = cur % 6; }
        g[0] = f2_one();
        g[1] = f2_pow(f2_xi(), e);
        for (int i = 2; i < 6; i++) g[i] = f2_mul(g[i - 1], g[1]);
        ready = true;
    }
    return g;
}
inline Fp12 f12_frob(const Fp12& a) {
    const Fp2* g = f12_frob_gammas();
    Fp12 r;
    for (int i = 0; i < 6; i++) r.c[i] = f2_mul(f2_conj(a.c[i]), g[i]);
    return r;
}
// a^2 for a in the cyclotomic subgroup (a^(p^6+1) = 1: everything after the easy part of the final exponentiation):
// Granger-Scott, 9 squarings in Fp2 instead of the 12 products of f12_sqr.  With x0..x5 = c0, c2, c4, c1, c3, c5:
//   (3 (x4^2 xi + x0^2) - 2 x0,  3 (x2^2 xi + x3^2) - 2 x1,  3 (x5^2 xi + x1^2) - 2 x2,
//    6 x1 x5 xi + 2 x3,          6 x0 x4 + 2 x4,              6 x2 x3 + 2 x5)
inline Fp12 f12_cyclo_sqr(const Fp12& a) {
    const Fp2 &x0 = a.c[0], &x1 = a.c[2], &x2 = a.c[4], &x3 = a.c[1], &x4 = a.c[3], &x5 = a.c[5];
    Fp2 t0 = f2_sqr(x4), t1 = f2_sqr(x0);
    const Fp2 t6 = f2_sub(f2_sub(f2_sqr(f2_add(x4, x0)), t0), t1);                 // 2 x4 x0
    Fp2 t2 = f2_sqr(x2), t3 = f2_sqr(x3);
    const Fp2 t7 = f2_sub(f2_sub(f2_sqr(f2_add(x2, x3)), t2), t3);                 // 2 x2 x3
    Fp2 t4 = f2_sqr(x5), t5 = f2_sqr(x1);
    const Fp2 t8 = f2_mul_xi(f2_sub(f2_sub(f2_sqr(f2_add(x5, x1)), t4), t5));      // 2 x5 x1 xi
    t0 = f2_add(f2_mul_xi(t0), t1);
    t2 = f2_add(f2_mul_xi(t2), t3);
    t4 = f2_add(f2_mul_xi(t4), t5);
    auto three_minus_two = [](const Fp2& t, const Fp2& x) { Fp2 d = f2_sub(t, x); return f2_add(f2_add(d, d), t); };   // 3t - 2x
    auto three_plus_two = [](const Fp2& t, const Fp2& x) { Fp2 d = f2_add(t, x); return f2_add(f2_add(d, d), t); };     // 3t + 2x
    Fp12 r;
    r.c[0] = three_minus_two(t0, x0);
    r.c[2] = three_minus_two(t2, x1);
    r.c[4] = three_minus_two(t4, x2);
    r.c[1] = three_plus_two(t8, x3);
    r.c[3] = three_plus_two(t6, x4);
    r.c[5] = three_plus_two(t7, x5);
    return r;
}
// a^x, x = 4965661367192848881 (the BN254 parameter); a in the cyclotomic subgroup
inline Fp12 f12_pow_x(const Fp12& a) {
    const uint64_t x = 4965661367192848881ull;
    Fp12 acc = a;
    for (int i = 61; i >= 0; i--) {   // bit 62 is the top bit of x
        acc = f12_cyclo_sqr(acc);
        if ((x >> i) & 1) acc = f12_mul(acc, a);
    }
    return acc;
}

// line through T and Q (twist points, T != -Q) evaluated at P = (xP, yP) in G1, then T <- T + Q
// l(P) = yP - lambda*xP * w + (lambda*xT - yT) * w^3     (psi(x',y') = (x' w^2, y' w^3))
inline Fp12 line_and_add(G2Affine* T, const G2Affine& Q, const Affine<Bn254Fp>& P, bool dbl) {
    Fp2 lam;
    if (dbl) {
        Fp2 xx = f2_sqr(T->x);
        lam = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(T->y, T->y)));
    } else {
        lam = f2_mul(f2_sub(Q.y, T->y), f2_inv(f2_sub(Q.x, T->x)));
    }
    Fp12 l;
    for (int i = 0; i < 6; i++) l.c[i] = f2_zero();
    l.c[0] = Fp2{P.y, fe_zero<Bn254Fp>()};
    l.c[1] = f2_neg(f2_mul_fp(lam, P.x));
    l.c[3] = f2_sub(f2_mul(lam, T->x), T->y);
    const G2Affine& O = dbl ? *T : Q;
    G2Affine R;
    R.x = f2_sub(f2_sub(f2_sqr(lam), T->x), O.x);
    R.y = f2_sub(f2_mul(lam, f2_sub(T->x, R.x)), T->y);
    R.inf = false;
    *T = R;
    return l;
}

// Inversion-free line steps: T = (X : Y : Z) projective on the twist (x = X/Z, y = Y/Z).  The line is scaled by a non-zero
// Fp2 factor (the cleared denominators), which the final exponentiation kills (Fp2 lies in a proper subfield).
struct G2Proj { Fp2 X, Y, Z; };
// tangent at T evaluated at P, then T <- 2T.   lambda = A/B, A = 3X^2, B = 2YZ;  line * (B Z)
inline Fp12 line_double_proj(G2Proj* T, const Affine<Bn254Fp>& P) {
    Fp2 XX = f2_sqr(T->X);
    Fp2 A = f2_add(f2_add(XX, XX), XX);
    Fp2 YZ = f2_mul(T->Y, T->Z);
    Fp2 B = f2_add(YZ, YZ);
    Fp2 BB = f2_sqr(B);
    Fp12 l;
    for (int i = 0; i < 6; i++) l.c[i] = f2_zero();
    l.c[0] = f2_mul_fp(f2_mul(B, T->Z), P.y);                               // B Z yP
    l.c[1] = f2_neg(f2_mul_fp(f2_mul(A, T->Z), P.x));                       // -A Z xP
    l.c[3] = f2_sub(f2_mul(A, T->X), f2_mul(B, T->Y));                      // A X - B Y
    Fp2 D = f2_mul(BB, T->Z);                                               // B^2 Z
    Fp2 XBB = f2_mul(T->X, BB);
    Fp2 N3 = f2_sub(f2_mul(f2_sqr(A), T->Z), f2_add(XBB, XBB));             // A^2 Z - 2 X B^2
    G2Proj R;
    R.X = f2_mul(N3, B);
    R.Y = f2_sub(f2_mul(A, f2_sub(XBB, N3)), f2_mul(T->Y, f2_mul(BB, B)));  // A (X B^2 - N3) - Y B^3
    R.Z = f2_mul(B, D);
    *T = R;
    return l;
}
// chord through T and the affine Q evaluated at P, then T <- T + Q.   lambda = A/B, A = y2 Z - Y, B = x2 Z - X;  line * B
inline Fp12 line_add_proj(G2Proj* T, const G2Affine& Q, const Affine<Bn254Fp>& P) {
    Fp2 A = f2_sub(f2_mul(Q.y, T->Z), T->Y);
    Fp2 B = f2_sub(f2_mul(Q.x, T->Z), T->X);
    Fp12 l;
    for (int i = 0; i < 6; i++) l.c[i] = f2_zero();
    l.c[0] = f2_mul_fp(B, P.y);
    l.c[1] = f2_neg(f2_mul_fp(A, P.x));
    l.c[3] = f2_sub(f2_mul(A, Q.x), f2_mul(B, Q.y));
    Fp2 BB = f2_sqr(B);
    Fp2 D = f2_mul(BB, T->Z);
    Fp2 N3 = f2_sub(f2_mul(f2_sqr(A), T->Z), f2_mul(BB, f2_add(T->X, f2_mul(Q.x, T->Z))));
    G2Proj R;
    R.X = f2_mul(N3, B);
    R.Y = f2_sub(f2_mul(A, f2_sub(f2_mul(Q.x, D), N3)), f2_mul(Q.y, f2_mul(B, D)));
    R.Z = f2_mul(B, D);
    *T = R;
    return l;
}
inline Fp12 miller_ate(const Affine<Bn254Fp>& P, const G2Affine& Q) {
    static const uint32_t S[4] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u};   // 6 x^2
    Fp12 f = f12_one();
    if (aff_is_inf<Bn254Fp>(P) || Q.inf) return f;
    G2Proj T{Q.x, Q.y, f2_one()};
    int top = 127;
    while (!((S[top >> 5] >> (top & 31)) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        f = f12_sqr(f);
        f = f12_mul_sparse(f, line_double_proj(&T, P));
        if ((S[i >> 5] >> (i & 31)) & 1) f = f12_mul_sparse(f, line_add_proj(&T, Q, P));
    }
    return f;
}

// The p-power Frobenius on the twist: psi^-1 o pi o psi (x', y') = (conj(x') gamma^2, conj(y') gamma^3), gamma = xi^((p-1)/6)
inline G2Affine g2_frob(const G2Affine& q) {
    const Fp2* g = f12_frob_gammas();
    return G2Affine{f2_mul(f2_conj(q.x), g[2]), f2_mul(f2_conj(q.y), g[3]), q.inf};
}
// OPTIMAL ate Miller function of the two pairs in ONE loop (the squaring of the running value is shared):
//   f_{6x+2,Q}(P) * l_{[6x+2]Q, pi(Q)}(P) * l_{[6x+2]Q + pi(Q), -pi^2(Q)}(P),   6x + 2 = 0x19d797039be763ba8 (65 bits)
// -- half the iterations of the plain ate function f_{6x^2,Q} (miller_ate, 127 bits), which stays as the form the slow reference
// is compared with.  Both are non-degenerate bilinear pairings, so the predicate "product == 1" is the same.
inline Fp12 miller_opt_ate2(const Affine<Bn254Fp>& P1, const G2Affine& Q1, const Affine<Bn254Fp>& P2, const G2Affine& Q2) {
    static const uint32_t S[3] = {0xbe763ba8u, 0x9d797039u, 0x1u};    // 6x + 2, little-endian words
    const bool on[2] = {!(aff_is_inf<Bn254Fp>(P1) || Q1.inf), !(aff_is_inf<Bn254Fp>(P2) || Q2.inf)};
    const Affine<Bn254Fp>* P[2] = {&P1, &P2};
    const G2Affine* Q[2] = {&Q1, &Q2};
    Fp12 f = f12_one();
    if (!on[0] && !on[1]) return f;
    G2Proj T[2] = {G2Proj{Q1.x, Q1.y, f2_one()}, G2Proj{Q2.x, Q2.y, f2_one()}};
    for (int i = 63; i >= 0; i--) {                      // bit 64 is the top bit
        f = f12_sqr(f);
        for (int k = 0; k < 2; k++) if (on[k]) f = f12_mul_sparse(f, line_double_proj(&T[k], *P[k]));
        if ((S[i >> 5] >> (i & 31)) & 1)
            for (int k = 0; k < 2; k++) if (on[k]) f = f12_mul_sparse(f, line_add_proj(&T[k], *Q[k], *P[k]));
    }
    for (int k = 0; k < 2; k++) {
        if (!on[k]) continue;
        const G2Affine q1 = g2_frob(*Q[k]);
        G2Affine q2 = g2_frob(q1);
        q2.y = f2_neg(q2.y);
        f = f12_mul_sparse(f, line_add_proj(&T[k], q1, *P[k]));
        f = f12_mul_sparse(f, line_add_proj(&T[k], q2, *P[k]));
    }
    return f;
}

// the same Miller function with affine steps (one Fp2 inversion each): the slow reference for the self-check
inline Fp12 miller_ate_affine(const Affine<Bn254Fp>& P, const G2Affine& Q) {
    // 6*x^2 = 147946756881789318990833708069417712966 = 0x6f4d8248eeb859fbf83e9682e87cfd46
    static const uint32_t S[4] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u};
    Fp12 f = f12_one();
    if (aff_is_inf<Bn254Fp>(P) || Q.inf) return f;
    G2Affine T = Q;
    int top = 127;
    while (!((S[top >> 5] >> (top & 31)) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        f = f12_mul(f, f);
        f = f12_mul(f, line_and_add(&T, T, P, true));
        if ((S[i >> 5] >> (i & 31)) & 1) {
            // T + Q never hits T == +-Q for Q of prime order r inside the loop (i*Q, i < r)
            f = f12_mul(f, line_and_add(&T, Q, P, false));
        }
    }
    return f;
}

// (p^12 - 1)/r as 32-bit little-endian limbs; generated by tools/gen_constants.py, checked in tests
#include "final_exp_limbs.inc"

inline Fp12 f12_pow_final(const Fp12& a) {
    Fp12 acc = f12_one();
    int top = FINAL_EXP_NLIMBS * 32 - 1;
    while (!((FINAL_EXP[top >> 5] >> (top & 31)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        acc = f12_mul(acc, acc);
        if ((FINAL_EXP[i >> 5] >> (i & 31)) & 1) acc = f12_mul(acc, a);
    }
    return acc;
}

// f^((p^12 - 1)/r): easy part by conjugation / inversion / Frobenius, hard part by the BN addition chain
// y0 * y1^2 * y2^6 * y3^12 * y4^18 * y5^30 * y6^36 (its exponent equals (p^4 - p^2 + 1)/r exactly)
inline Fp12 f12_final_exp(const Fp12& a) {
    Fp12 f = f12_mul(f12_conj(a), f12_inv(a));          // a^(p^6 - 1)
    f = f12_mul(f12_frob(f12_frob(f)), f);              // ^(p^2 + 1): now in the cyclotomic subgroup, f^-1 = conj(f)
    Fp12 fx = f12_pow_x(f), fx2 = f12_pow_x(fx), fx3 = f12_pow_x(fx2);
    Fp12 fp = f12_frob(f), fp2 = f12_frob(fp), fp3 = f12_frob(fp2);
    Fp12 y0 = f12_mul(f12_mul(fp, fp2), fp3);
    Fp12 y1 = f12_conj(f);
    Fp12 y2 = f12_frob(f12_frob(fx2));
    Fp12 y3 = f12_conj(f12_frob(fx));
    Fp12 y4 = f12_conj(f12_mul(fx, f12_frob(fx2)));
    Fp12 y5 = f12_conj(fx2);
    Fp12 y6 = f12_conj(f12_mul(fx3, f12_frob(fx3)));
    Fp12 t0 = f12_mul(f12_mul(f12_sqr(y6), y4), y5);
    Fp12 t1 = f12_mul(f12_mul(y3, y5), t0);
    t0 = f12_mul(t0, y2);
    t1 = f12_mul(f12_sqr(t1), t0);
    t1 = f12_sqr(t1);
    t0 = f12_mul(t1, y1);
    t1 = f12_mul(t1, y0);
    t0 = f12_sqr(t0);
    return f12_mul(t0, t1);
}

// e(P1,Q1) * e(P2,Q2) == 1 ?   slow = true: affine Miller steps + the literal exponent (the reference form)
inline bool pairing_product_is_one(const Affine<Bn254Fp>& P1, const G2Affine& Q1, const Affine<Bn254Fp>& P2,
                                   const G2Affine& Q2, bool slow = false) {
    if (slow) return f12_is_one(f12_pow_final(f12_mul(miller_ate_affine(P1, Q1), miller_ate_affine(P2, Q2))));
    return f12_is_one(f12_final_exp(miller_opt_ate2(P1, Q1, P2, Q2)));
}

}  // namespace porla
