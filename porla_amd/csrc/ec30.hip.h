// Mixed addition of the bucket accumulation in the reduced-radix field form of fe30.hip.h (30-bit limbs, radix 2^270).
// Same group law and exceptional cases as ec.hip.h:xyzz_madd (the reference: gnark-crypto's g1JacExtended.addMixed behind
// G1Affine.MultiExp, porla/main.go:136) -- only the residues' representation differs, so the bucket sums are the same
// group elements and the MSM result stays bit-exact.
//
// Value bounds (multiples of p; a product's result is < p + 2^246, written "1"):
//   X1 <= 5, Y1 <= 4 (the accumulator: Y3 <= 3 below, 4p - Y <= 4 after xyzz30_flip_finish or a negation), ZZ1, ZZZ1 <= 1, X2, Y2 < 1
//   U2 = X2 ZZ1, S2 = Y2 ZZZ1                      <= 1
//   P  = U2 - X1 + 6p                              <= 7
//   Rn = Y1 - S2 + 2p = -R                         <= 6   (the sign-alternating form below; Y1 is the minuend, so its 4p needs no allowance)
//   PP = P^2, PPP = P PP, Q = X1 PP, RR = Rn^2     <= 1
//   E  = PPP + 2 Q                                 <= 3
//   X3 = RR - E + 4p                               <= 5
//   D  = Q - X3 + 6p                               <= 7
//   Y3 = R D - Y1 PPP                              <= 1 as one two-product reduction (-Y3 = Rn D + Y1 PPP), <= 3 as T1 - T2 + 2p in the full addition
// Everything is below 8p < 2^257, so limb 8 stays below 2^18 as the products require.
#pragma once
#include "ec.hip.h"
#include "fe30.hip.h"

namespace porla {

template <class M>
struct XYZZ30 {
    F30<M> x, y, zz, zzz;
    bool inf;
};

template <class M>
__device__ __forceinline__ XYZZ<M> xyzz30_to_xyzz(const XYZZ30<M>& p);   // defined at the end of this file

// 2 * (affine a) in the 2^270 form; a != infinity.  Rare (a bucket receives the point it already holds): kept out of line.
// Arguments by value (registers): a pointer to the caller's copies would make the compiler spill them on every iteration.
template <class M>
__device__ __noinline__ XYZZ30<M> xyzz30_double_affine(F30<M> ax, F30<M> ay) {
    XYZZ30<M> r;
    // mdbl-2008-s-1: U = 2Y, V = U^2, W = U V, S = X V, M = 3 X^2, X3 = M^2 - 2S, Y3 = M (S - X3) - W Y, ZZ3 = V, ZZZ3 = W
    F30<M> U = f30_small_mul<M, 2>(ay);
    F30<M> V = f30_sqr<M>(U);
    F30<M> W = f30_mul<M>(U, V);
    F30<M> S = f30_mul<M>(ax, V);
    F30<M> XX = f30_sqr<M>(ax);
    F30<M> Mm = f30_small_mul<M, 3>(XX);            // <= 3
    F30<M> MM = f30_sqr<M>(Mm);
    F30<M> S2 = f30_small_mul<M, 2>(S);             // <= 2
    F30<M> X3 = f30_sub<M, 3>(MM, S2);              // <= 4
    F30<M> D = f30_sub<M, 5>(S, X3);                // <= 6
    F30<M> T1 = f30_mul<M>(Mm, D);
    F30<M> T2 = f30_mul<M>(W, ay);
    r.x = X3;
    r.y = f30_sub<M, 2>(T1, T2);                    // <= 3
    r.zz = V;
    r.zzz = W;
    r.inf = f30_product_is_zero<M>(V);              // y = 0: a point of order 2 (none on these curves)
    return r;
}

// The mixed addition p += a (a affine in the 2^270 form) for a RUN of additions into one accumulator (the bucket sums, the rows of a
// fixed-base commitment), with
// Y3 = R D - Y1 PPP as ONE two-product reduction (f30_mul2_mont) and no subtraction behind it:
//   with Rn = Y1 - S2 = -R the sum  Rn D + Y1 PPP  is -Y3, so (X3, Rn D + Y1 PPP, ZZ3, ZZZ3) is a valid representation of
//   -(p + a).  The accumulator therefore changes sign with every addition; `flip` says whether it currently holds the negative of
//   the true sum, the caller negates the incoming point when it does (that negation is folded into the digit's own sign:
//   xyzz30_flip_neg), and xyzz30_flip_finish undoes an odd count at the end.  Same group elements as xyzz30_madd, 154 instructions
//   fewer per addition (one Montgomery reduction and one subtraction-with-ripple).
// Bounds: as xyzz30_madd, with Rn = Y1 - S2 + 2p <= 6 (Y1 <= 4 when the accumulator was loaded from memory -- a sum this form
// left there after an odd count: 4p - Y -- and <= 1 after an addition here) and Y3 <= 1; every operand of a product stays far
// below 2^258.  Both field forms have the two-product reduction (f30_mul2: Montgomery f30_mul2_mont, special form
// f30_mul2_pm with one fold).
template <class M>
__device__ __forceinline__ bool xyzz30_flip_neg(bool digit_negative, bool flip) { return digit_negative != flip; }
template <class M>
__device__ __forceinline__ void xyzz30_madd_flip(XYZZ30<M>& p, bool& flip, const F30<M>& ax, const F30<M>& ay) {
    if (p.inf) {                                        // the first point: the sum is (flip ? -a : a) as the caller negated it
        p.x = ax; p.y = ay;
        p.zz = f30_const<M>(M::R1_30); p.zzz = p.zz;
        p.inf = false;
        return;
    }
    F30<M> U2 = f30_mul<M>(ax, p.zz);
    F30<M> S2 = f30_mul<M>(ay, p.zzz);
    F30<M> Pp = f30_sub<M, 6>(U2, p.x);
    F30<M> Rn = f30_sub<M, 2>(p.y, S2);
    F30<M> PP = f30_sqr<M>(Pp);
    if (f30_product_is_zero<M>(PP)) {                   // same x: the same point (double it) or its negative (infinity); no flip
        F30<M> RR = f30_sqr<M>(Rn);
        if (f30_product_is_zero<M>(RR)) p = xyzz30_double_affine<M>(ax, ay);
        else p.inf = true;
        return;
    }
    F30<M> PPP = f30_mul<M>(Pp, PP);
    F30<M> Q = f30_mul<M>(p.x, PP);
    F30<M> RR = f30_sqr<M>(Rn);
    F30<M> E = f30_add2<M>(PPP, Q);
    F30<M> X3 = f30_sub<M, 4>(RR, E);
    F30<M> D = f30_sub<M, 6>(Q, X3);
    p.y = f30_mul2<M>(Rn, D, p.y, PPP);                 // -Y3
    p.x = X3;
    p.zz = f30_mul<M>(p.zz, PP);
    p.zzz = f30_mul<M>(p.zzz, PPP);
    flip = !flip;
}
// The form the accumulation loops call FIRST: the addition above as straight-line code.  Returns false -- with p and flip untouched --
// when this lane meets an exceptional case (accumulator or point at infinity, equal x); the caller then leaves its fast loop and
// finishes the lane's entries with the general form xyzz30_madd_flip.  With the exceptional cases as branches INSIDE the hot loop
// the compiler staged the whole accumulator through a second register set at their joins (~42 copies per addition on the common
// path); a loop exit has no join per iteration.  Both forms leave the same group element under the same sign convention.
template <class M>
__device__ __forceinline__ bool xyzz30_madd_flip_fast(XYZZ30<M>& p, bool& flip, const F30<M>& ax, const F30<M>& ay, bool a_is_inf) {
    // (an accumulator at infinity holds zeros: the products below are then meaningless and unused)
    F30<M> U2 = f30_mul<M>(ax, p.zz);
    F30<M> S2 = f30_mul<M>(ay, p.zzz);
    F30<M> Pp = f30_sub<M, 6>(U2, p.x);
    F30<M> Rn = f30_sub<M, 2>(p.y, S2);
    F30<M> PP = f30_sqr<M>(Pp);
    if (a_is_inf || p.inf || f30_product_is_zero<M>(PP)) return false;
    F30<M> PPP = f30_mul<M>(Pp, PP);
    F30<M> Q = f30_mul<M>(p.x, PP);
    F30<M> RR = f30_sqr<M>(Rn);
    F30<M> E = f30_add2<M>(PPP, Q);
    F30<M> X3 = f30_sub<M, 4>(RR, E);
    F30<M> D = f30_sub<M, 6>(Q, X3);
    p.y = f30_mul2<M>(Rn, D, p.y, PPP);             // -Y3
    p.x = X3;
    p.zz = f30_mul<M>(p.zz, PP);
    p.zzz = f30_mul<M>(p.zzz, PPP);
    flip = !flip;
    return true;
}
// The same step for an accumulator that IS an affine point (just copied in: ZZ = ZZZ = the unit): U2 = X2, S2 = Y2, ZZ3 = PP, ZZZ3 = PPP --
// the four products with the unit drop out (4M + 2S instead of 8M + 2S; mmadd-2008-s).  One addition per work item: with ~16 entries
// per bucket that is a fifteenth of all additions at 45 % less.  Bounds: X2, Y2 canonical (<= 1) stand where U2, S2 (<= 1) stood.
// Returns false, p untouched, in the exceptional cases.
template <class M>
__device__ __forceinline__ bool xyzz30_mmadd_flip_fast(XYZZ30<M>& p, bool& flip, const F30<M>& ax, const F30<M>& ay, bool a_is_inf) {
    F30<M> Pp = f30_sub<M, 6>(ax, p.x);
    F30<M> Rn = f30_sub<M, 2>(p.y, ay);
    F30<M> PP = f30_sqr<M>(Pp);
    if (a_is_inf || p.inf || f30_product_is_zero<M>(PP)) return false;
    F30<M> PPP = f30_mul<M>(Pp, PP);
    F30<M> Q = f30_mul<M>(p.x, PP);
    F30<M> RR = f30_sqr<M>(Rn);
    F30<M> E = f30_add2<M>(PPP, Q);
    F30<M> X3 = f30_sub<M, 4>(RR, E);
    F30<M> D = f30_sub<M, 6>(Q, X3);
    p.y = f30_mul2<M>(Rn, D, p.y, PPP);             // -Y3
    p.x = X3;
    p.zz = PP;
    p.zzz = PPP;
    flip = !flip;
    return true;
}
template <class M>
__device__ __forceinline__ XYZZ30<M> xyzz30_infinity() {
    XYZZ30<M> p;
#pragma unroll
    for (int i = 0; i < 9; i++) { p.x.v[i] = 0; p.y.v[i] = 0; p.zz.v[i] = 0; p.zzz.v[i] = 0; }
    p.inf = true;
    return p;
}
// the true sum: Y -> 4p - Y (<= 4p) when the accumulator holds its negative.  BN254: 4p < 2^256, storable as it is in the lazy memory
// form; secp256k1 (4p > 2^256): xyzz30_store_lazy reduces every residue canonically on the way out, so the bound never reaches memory.
// Consumers: 2 Y <= 8p in a doubling -- inside the products' budgets (tools/check_fe30_bounds.py:check_value_ranges)
template <class M>
__device__ __forceinline__ void xyzz30_flip_finish(XYZZ30<M>& p, bool flip) {
    if (flip && !p.inf) p.y = f30_sub<M, 4>(F30<M>{}, p.y);
}

// 2 * p (dbl-2008-s-1, a = 0); p not infinity.  Out of line, operands by value (see xyzz30_double_affine).
//   U = 2 Y1 (<= 6), V = U^2, W = U V, S = X1 V, M = 3 X1^2 (<= 3), X3 = M^2 - 2S + 3p (<= 4), Y3 = M (S - X3 + 5p) - W Y1 + 2p (<= 3)
template <class M>
__device__ __forceinline__ XYZZ30<M> xyzz30_double_body(const F30<M>& x, const F30<M>& y, const F30<M>& zz, const F30<M>& zzz) {
    XYZZ30<M> r;
    F30<M> U = f30_small_mul<M, 2>(y);
    F30<M> V = f30_sqr<M>(U);
    F30<M> W = f30_mul<M>(U, V);
    F30<M> S = f30_mul<M>(x, V);
    F30<M> XX = f30_sqr<M>(x);
    F30<M> Mm = f30_small_mul<M, 3>(XX);
    F30<M> MM = f30_sqr<M>(Mm);
    F30<M> S2 = f30_small_mul<M, 2>(S);
    F30<M> X3 = f30_sub<M, 3>(MM, S2);
    F30<M> D = f30_sub<M, 5>(S, X3);
    F30<M> T1 = f30_mul<M>(Mm, D);
    F30<M> T2 = f30_mul<M>(W, y);
    r.x = X3;
    r.y = f30_sub<M, 2>(T1, T2);
    r.zz = f30_mul<M>(V, zz);
    r.zzz = f30_mul<M>(W, zzz);
    r.inf = f30_product_is_zero<M>(V);
    return r;
}
template <class M>
__device__ __noinline__ XYZZ30<M> xyzz30_double(F30<M> x, F30<M> y, F30<M> zz, F30<M> zzz) {
    return xyzz30_double_body<M>(x, y, zz, zzz);
}

// p += q (add-2008-s), all exceptional cases.  Bounds as for the mixed form: X <= 5, Y <= 4 (products only), ZZ, ZZZ <= 1 on both sides;
//   U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1 (<= 1);  P = U2 - U1 + 2p (<= 3);  R = S2 - S1 + 2p (<= 3);
//   Q = U1 PP;  X3 = RR - (PPP + 2Q) + 4p (<= 5);  Y3 = R (Q - X3 + 6p) - S1 PPP + 2p (<= 3)
template <class M>
__device__ __forceinline__ void xyzz30_add(XYZZ30<M>& p, const XYZZ30<M>& q) {
    if (q.inf) return;
    if (p.inf) { p = q; return; }
    F30<M> U1 = f30_mul<M>(p.x, q.zz);
    F30<M> U2 = f30_mul<M>(q.x, p.zz);
    F30<M> S1 = f30_mul<M>(p.y, q.zzz);
    F30<M> S2 = f30_mul<M>(q.y, p.zzz);
    F30<M> Pp = f30_sub<M, 2>(U2, U1);
    F30<M> Rr = f30_sub<M, 2>(S2, S1);
    F30<M> PP = f30_sqr<M>(Pp);
    F30<M> RR = f30_sqr<M>(Rr);
    if (f30_product_is_zero<M>(PP)) {
        if (f30_product_is_zero<M>(RR)) p = xyzz30_double<M>(p.x, p.y, p.zz, p.zzz);
        else p.inf = true;
        return;
    }
    F30<M> PPP = f30_mul<M>(Pp, PP);
    F30<M> Q = f30_mul<M>(U1, PP);
    F30<M> E = f30_add2<M>(PPP, Q);
    F30<M> X3 = f30_sub<M, 4>(RR, E);
    F30<M> D = f30_sub<M, 6>(Q, X3);
    F30<M> T1 = f30_mul<M>(Rr, D);
    F30<M> T2 = f30_mul<M>(S1, PPP);
    p.x = X3;
    p.y = f30_sub<M, 2>(T1, T2);
    p.zz = f30_mul<M>(f30_mul<M>(p.zz, q.zz), PP);
    p.zzz = f30_mul<M>(f30_mul<M>(p.zzz, q.zzz), PPP);
}

// Memory form between the kernels of the reduced-radix path ("lazy"): BN254 -- the four residues in the 2^270 Montgomery
// form, NOT reduced (X <= 5p + 2^246 < 2^256 and the others below that), each packed into 8 words; secp256k1 (p ~ 2^256) --
// the four residues reduced to canonical form on the way out.  Infinity = all words zero (a finite point's ZZ is never
// 0 mod p; BN254: a stored ZZ is a product's result below p + 2^246 that is not p, so never the word pattern 0).
__device__ __forceinline__ void store_words8(uint32_t* d, const uint32_t (&w)[8]) {
    uint4* q = reinterpret_cast<uint4*>(d);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
template <class M>
__device__ __forceinline__ void xyzz30_store_lazy(XYZZ<M>* dst, const XYZZ30<M>& p) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    if (p.inf) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4* q = reinterpret_cast<uint4*>(d);
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = z;
        return;
    }
    Fe<M> t;
    if constexpr (M::PSEUDO_MERSENNE) {
        // special-form modulus (p ~ 2^256): an unreduced X <= 5p does not fit 256 bits and a product's result may have bit 256
        // set, so the four residues are stored canonically here (one fold + at most two subtractions each)
        t = f30_to_fe_canonical<M>(f30_pm_reduce<M>(p.x));   store_words8(d, t.v);
        t = f30_to_fe_canonical<M>(f30_pm_reduce<M>(p.y));   store_words8(d + 8, t.v);
        t = f30_to_fe_canonical<M>(p.zz);                    store_words8(d + 16, t.v);
        t = f30_to_fe_canonical<M>(p.zzz);                   store_words8(d + 24, t.v);
    } else {
        f30_pack<M>(t.v, p.x);   store_words8(d, t.v);
        f30_pack<M>(t.v, p.y);   store_words8(d + 8, t.v);
        f30_pack<M>(t.v, p.zz);  store_words8(d + 16, t.v);
        f30_pack<M>(t.v, p.zzz); store_words8(d + 24, t.v);
    }
}
template <class M>
__device__ __forceinline__ XYZZ30<M> xyzz30_load_lazy(const XYZZ<M>* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
    uint4 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = q[i];
    XYZZ30<M> p;
    uint32_t t[8];
    t[0] = w[0].x; t[1] = w[0].y; t[2] = w[0].z; t[3] = w[0].w; t[4] = w[1].x; t[5] = w[1].y; t[6] = w[1].z; t[7] = w[1].w;
    p.x = f30_unpack<M>(t);
    t[0] = w[2].x; t[1] = w[2].y; t[2] = w[2].z; t[3] = w[2].w; t[4] = w[3].x; t[5] = w[3].y; t[6] = w[3].z; t[7] = w[3].w;
    p.y = f30_unpack<M>(t);
    t[0] = w[4].x; t[1] = w[4].y; t[2] = w[4].z; t[3] = w[4].w; t[4] = w[5].x; t[5] = w[5].y; t[6] = w[5].z; t[7] = w[5].w;
    p.zz = f30_unpack<M>(t);
    p.inf = (t[0] | t[1] | t[2] | t[3] | t[4] | t[5] | t[6] | t[7]) == 0;
    t[0] = w[6].x; t[1] = w[6].y; t[2] = w[6].z; t[3] = w[6].w; t[4] = w[7].x; t[5] = w[7].y; t[6] = w[7].z; t[7] = w[7].w;
    p.zzz = f30_unpack<M>(t);
    return p;
}

// Out-of-line group operations on the lazy memory form (for the scalar-multiplication ladders of mac_fft.hip.h, whose
// working set -- a table of multiples -- lives in private memory anyway): one body of each in the instruction cache.
//   *p += (neg ? -1 : 1) * (phi ? (beta X, Y, ZZ, ZZZ) : (X, Y, ZZ, ZZZ)) of *q;   beta30 = beta in the 2^270 form
template <class M>
__device__ __noinline__ void xyzz30_add_mem(XYZZ<M>* p, const XYZZ<M>* q, uint32_t neg, uint32_t phi, const F30<M>* beta30) {
    XYZZ30<M> a = xyzz30_load_lazy<M>(p), b = xyzz30_load_lazy<M>(q);
    if (!b.inf) {
        if (phi) b.x = f30_mul<M>(b.x, *beta30);
        if (neg) b.y = f30_sub<M, 4>(F30<M>{}, b.y);       // 4p - Y <= 4p: fine as an operand, never stored
    }
    xyzz30_add<M>(a, b);
    xyzz30_store_lazy<M>(p, a);
}
template <class M>
__device__ __noinline__ void xyzz30_double_mem(XYZZ<M>* p, int times) {
    XYZZ30<M> a = xyzz30_load_lazy<M>(p);
    if (a.inf) return;
#pragma unroll 1
    for (int t = 0; t < times && !a.inf; t++) a = xyzz30_double_body<M>(a.x, a.y, a.zz, a.zzz);
    xyzz30_store_lazy<M>(p, a);
}

// ---------------------------------------------------------------- one addition on the four lanes of a quad
// The small levels of the reduction tree are chains of dependent additions with nothing else to run: a lone wave pays
// ~0.5 us per field product, 14 of them in sequence.  Here the 14 products of p + q are spread over 4 lanes in 4 rounds
// (values travel between the lanes by DPP quad permutes), so the chain is 4 products long:
//   lane          0              1               2                3
//   loads     X1, ZZ2        X2, ZZ1         Y1, ZZZ2         Y2, ZZZ1
//   round 1   U1 = X1 ZZ2    U2 = X2 ZZ1     S1 = Y1 ZZZ2     S2 = Y2 ZZZ1
//   round 2   PP = (U2-U1)^2 ZZ1 ZZ2         ZZZ1 ZZZ2        RR = (S2-S1)^2
//   round 3   PPP = P PP     ZZ3 = . PP      -                Q = U1 PP
//   round 4   T2 = S1 PPP    -               ZZZ3 = . PPP     T1 = R (Q - X3),  X3 = RR - PPP - 2Q
//   result                   ZZ3             ZZZ3             X3,  Y3 = T1 - T2
// Exceptional operands (infinity, p = +-q) are detected in rounds 1-2 and handed to one lane's ordinary addition.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <class M, int CTRL>
__device__ __forceinline__ F30<M> f30_quad(const F30<M>& a) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = dpp_quad<CTRL>(a.v[i]);
    return r;
}
template <class M>
__device__ __forceinline__ F30<M> f30_sel(bool c, const F30<M>& a, const F30<M>& b) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
// one coordinate (0 = X, 1 = Y, 2 = ZZ, 3 = ZZZ) of a memory-form point
template <class M>
__device__ __forceinline__ F30<M> xyzz30_load_coord(const XYZZ<M>* p, int c, bool* all_zero) {
    const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(p) + 8 * c);
    const uint4 a = q[0], b = q[1];
    uint32_t t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *all_zero = (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) == 0;
    return f30_unpack<M>(t);
}
// final: the 2^256 Montgomery form the host reads; otherwise the lazy memory form
template <class M>
__device__ __forceinline__ void xyzz30_store_coord(XYZZ<M>* p, int c, const F30<M>& v, bool final) {
    uint32_t* d = reinterpret_cast<uint32_t*>(p) + 8 * c;
    Fe<M> t;
    if (final) {
        t = f30_to_fe_canonical<M>(f30_mul<M>(v, f30_const<M>(M::R1)));
    } else if constexpr (M::PSEUDO_MERSENNE) {
        t = f30_to_fe_canonical<M>(f30_pm_reduce<M>(v));
    } else {
        f30_pack<M>(t.v, v);
    }
    store_words8(d, t.v);
}
template <class M>
__device__ __noinline__ void xyzz30_add_one_lane(const XYZZ<M>* pa, const XYZZ<M>* pb, XYZZ<M>* out, bool final) {
    XYZZ30<M> a = xyzz30_load_lazy<M>(pa), b = xyzz30_load_lazy<M>(pb);
    xyzz30_add<M>(a, b);
    if (final) {
        const XYZZ<M> r = xyzz30_to_xyzz<M>(a);
        uint32_t* d = reinterpret_cast<uint32_t*>(out);
        store_words8(d, r.x.v); store_words8(d + 8, r.y.v); store_words8(d + 16, r.zz.v); store_words8(d + 24, r.zzz.v);
    } else {
        xyzz30_store_lazy<M>(out, a);
    }
}
// *out = *pa + *pb.  Called by ALL FOUR lanes of a quad with the same arguments; lane = position in the wave;
// live = false: compute but do not store (padding quads keep the permutes well defined).
template <class M>
__device__ __forceinline__ void xyzz30_add_quad(const XYZZ<M>* pa, const XYZZ<M>* pb, XYZZ<M>* out, bool final, bool live,
                                                uint32_t lane) {
    const uint32_t r = lane & 3u;
    const XYZZ<M>* sa = (r & 1u) ? pb : pa;
    const XYZZ<M>* sb = (r & 1u) ? pa : pb;
    bool za, zb;
    const F30<M> A = xyzz30_load_coord<M>(sa, (int)(r >> 1), &za);
    const F30<M> B = xyzz30_load_coord<M>(sb, 2 + (int)(r >> 1), &zb);
    bool special = zb;                                   // a ZZ / ZZZ of zero: that operand is infinity
    if (!final) {
        // an infinite operand (common while the sums are sparse: small inputs, the first tree levels): the result is the other
        // operand, copied coordinate by coordinate in the memory form -- no arithmetic, and not the one-lane path below
        const uint32_t qz = (uint32_t)(__ballot(zb) >> (lane & 60u)) & 0xfu;   // even lanes looked at *pb, odd lanes at *pa
        if (qz) {
            const XYZZ<M>* src = (qz & 0x5u) ? pa : pb;
            if (live && src != out) {
                const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(src) + 8 * r);
                uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(out) + 8 * r);
                const uint4 v0 = q[0], v1 = q[1];
                d[0] = v0; d[1] = v1;
            }
            return;
        }
    }
    const F30<M> M1 = f30_mul<M>(A, B);                  // U1, U2, S1, S2
    const bool edge = (r == 0u) || (r == 3u);
    const F30<M> rcv = f30_quad<M, 0xB1>(f30_sel<M>(edge, B, M1));   // lanes 0<->1, 2<->3
    const F30<M> D = f30_sub<M, 2>(f30_sel<M>(r == 0u, rcv, M1), f30_sel<M>(r == 0u, M1, rcv));   // lane 0: P, lane 3: R
    const F30<M> M2 = f30_mul<M>(f30_sel<M>(edge, D, B), f30_sel<M>(edge, D, rcv));           // PP, ZZ1 ZZ2, ZZZ1 ZZZ2, RR
    if (r == 0u && f30_product_is_zero<M>(M2)) special = true;                                    // same x
    const unsigned long long bal = __ballot(special);
    if ((bal >> (lane & 60u)) & 0xfull) {
        if (r == 0u && live) xyzz30_add_one_lane<M>(pa, pb, out, final);
        return;
    }
    const F30<M> bPP = f30_quad<M, 0x00>(M2);
    const F30<M> bU1 = f30_quad<M, 0x00>(M1);
    const F30<M> M3 = f30_mul<M>(r == 0u ? D : (r == 3u ? bU1 : M2), r == 0u ? M2 : bPP);       // PPP, ZZ3, -, Q
    const F30<M> bPPP = f30_quad<M, 0x00>(M3);
    const F30<M> bS1 = f30_quad<M, 0xAA>(M1);
    const F30<M> E = f30_add2<M>(bPPP, M3);              // lane 3: PPP + 2Q
    const F30<M> X3 = f30_sub<M, 4>(M2, E);              // lane 3: RR - E
    const F30<M> Dq = f30_sub<M, 6>(M3, X3);             // lane 3: Q - X3
    const F30<M> M4 = f30_mul<M>(r == 0u ? bS1 : (r == 3u ? D : M2), r == 3u ? Dq : bPPP);       // T2, -, ZZZ3, T1
    const F30<M> bT2 = f30_quad<M, 0x00>(M4);
    const F30<M> Y3 = f30_sub<M, 2>(M4, bT2);            // lane 3
    if (!live) return;
    if (r == 3u) { xyzz30_store_coord<M>(out, 0, X3, final); xyzz30_store_coord<M>(out, 1, Y3, final); }
    else if (r == 1u) xyzz30_store_coord<M>(out, 2, M3, final);
    else if (r == 2u) xyzz30_store_coord<M>(out, 3, M4, final);
}

// *out = 2 * *p on the four lanes of a quad (dbl-2008-s-1, a = 0; bounds as xyzz30_double_body): the 9 products in 3 rounds.
//   lane          0                    1                  2               3
//   loads     X                    Y                  ZZ              ZZZ
//   round 1   XX = X^2             V = (2Y)^2         -               -
//   round 2   S = X V              W = 2Y V           ZZ3 = ZZ V      MM = (3 XX)^2
//   round 3   T1 = 3XX (S - X3)    T2 = W Y           -               ZZZ3 = ZZZ W       X3 = MM - 2S
//   result    X3, Y3 = T1 - T2                        ZZ3             ZZZ3
// Called by ALL FOUR lanes of a quad with the same arguments (out may be p); infinity stays infinity.
template <class M>
__device__ __forceinline__ void xyzz30_dbl_quad(const XYZZ<M>* p, XYZZ<M>* out, bool live, uint32_t lane) {
    const uint32_t r = lane & 3u;
    bool zero;
    const F30<M> A = xyzz30_load_coord<M>(p, (int)r, &zero);
    const uint32_t qz = (uint32_t)(__ballot(zero && r == 2u) >> (lane & 60u)) & 0xfu;      // ZZ = 0: infinity
    if (qz) {
        if (live && out != p) {
            uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(out) + 8 * r);
            d[0] = make_uint4(0, 0, 0, 0); d[1] = d[0];
        }
        return;
    }
    const F30<M> U = f30_small_mul<M, 2>(A);                               // lane 1: 2Y
    const F30<M> O1 = f30_sel<M>(r == 1u, U, A);
    const F30<M> M1 = f30_mul<M>(O1, O1);                                  // XX, V, -, -
    const F30<M> bV = f30_quad<M, 0x55>(M1);
    const F30<M> bXX = f30_quad<M, 0x00>(M1);
    const F30<M> Mm = f30_small_mul<M, 3>(bXX);                            // 3 X^2 (lanes 0 and 3 use it)
    const F30<M> M2 = f30_mul<M>(r == 3u ? Mm : O1, r == 3u ? Mm : bV);    // S, W, ZZ3, MM
    const F30<M> bW = f30_quad<M, 0x55>(M2);
    const F30<M> bMM = f30_quad<M, 0xFF>(M2);
    const F30<M> S2 = f30_small_mul<M, 2>(M2);                             // lane 0: 2S
    const F30<M> X3 = f30_sub<M, 3>(bMM, S2);                              // lane 0
    const F30<M> D = f30_sub<M, 5>(M2, X3);                                // lane 0: S - X3
    const F30<M> M3 = f30_mul<M>(r == 0u ? Mm : bW, r == 0u ? D : A);      // T1, T2 = W Y, (W ZZ), ZZZ3 = W ZZZ
    const F30<M> bT2 = f30_quad<M, 0x55>(M3);
    const F30<M> Y3 = f30_sub<M, 2>(M3, bT2);                              // lane 0
    if (!live) return;
    if (r == 0u) { xyzz30_store_coord<M>(out, 0, X3, false); xyzz30_store_coord<M>(out, 1, Y3, false); }
    else if (r == 2u) xyzz30_store_coord<M>(out, 2, M2, false);
    else if (r == 3u) xyzz30_store_coord<M>(out, 3, M3, false);
}

// ---------------------------------------------------------------- the same two operations with the point IN REGISTERS
// A scalar-multiplication ladder (mac_fft.hip.h) is ~200 dependent group operations on ONE accumulator: through memory every
// operation paid a pack + store + barrier + load + unpack of the accumulator (and, for one wave per SIMD, the exposed LDS round
// trip).  Here the accumulator stays distributed over the quad's registers between operations -- lane r holds coordinate r
// (0 = X, 1 = Y, 2 = ZZ, 3 = ZZZ) as an F30 -- and only the OTHER operand (a table entry) comes from memory, one coordinate per
// lane.  Same formulas, same bounds: X <= 5p, Y <= 4p, ZZ / ZZZ product results; nothing is reduced between operations
// (8p < 2^258 for BN254, < 2^259 for secp256k1: tools/check_fe30_bounds.py).  Infinity is the caller's flag, not a value.
//
// c = 2 c.  Called by all four lanes of a quad whose accumulator is not infinity.  The lane roles are those of xyzz30_dbl_quad,
// with Y3 formed on lane 1 (it holds T2 and receives T1), so the result lands where the next operation expects it: no permute.
template <class M>
__device__ __forceinline__ void xyzz30_dbl_quadreg(F30<M>& c, uint32_t r) {
    F30<M> O1;                                                             // lane 1: 2Y (a shift by the lane's own amount), else c
    {
        const uint32_t sh = r == 1u ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < 9; i++) O1.v[i] = c.v[i] << sh;
        f30_ripple<M>(O1);
    }
    const F30<M> M1 = f30_sqr<M>(O1);                                      // XX, V, -, -
    const F30<M> bV = f30_quad<M, 0x55>(M1);
    const F30<M> bXX = f30_quad<M, 0x00>(M1);
    const F30<M> Mm = f30_small_mul<M, 3>(bXX);                            // 3 X^2 (lanes 0 and 3 use it)
    const F30<M> M2 = f30_mul<M>(r == 3u ? Mm : O1, r == 3u ? Mm : bV);    // S, W, ZZ3, MM
    const F30<M> bW = f30_quad<M, 0x55>(M2);
    const F30<M> bMM = f30_quad<M, 0xFF>(M2);
    const F30<M> X3 = f30_sub_twice<M, 3>(bMM, M2);                        // lane 0: MM - 2S
    const F30<M> D = f30_sub<M, 5>(M2, X3);                                // lane 0: S - X3
    const F30<M> M3 = f30_mul<M>(r == 0u ? Mm : bW, r == 0u ? D : c);      // T1, T2 = W Y, (W ZZ), ZZZ3 = W ZZZ
    const F30<M> bT1 = f30_quad<M, 0x00>(M3);
    const F30<M> Y3 = f30_sub<M, 2>(bT1, M3);                              // lane 1: T1 - T2
    c = r == 0u ? X3 : (r == 1u ? Y3 : (r == 2u ? M2 : M3));
}
// c = c + (neg ? -1 : 1) * q, q a memory-form point (neither operand infinity -- the caller's flags); `qx`: where q's X is read
// from (the table of X scaled by beta for the endomorphism's half, else q itself).  Lane roles (a, b, c, d of xyzz30_add_quad's
// table) sit on lanes 1, 2, 3, 0, so that X3, Y3, ZZ3, ZZZ3 come out on lanes 0, 1, 2, 3: ONE permute of the accumulator on the
// way in (lane 0 <- ZZZ1, lane 1 <- X1, lane 3 <- Y1), none on the way out; each lane loads one coordinate of q:
//   lane          1 (a)          2 (b)            3 (c)             0 (d)
//   operands  X1, ZZ2        X2, ZZ1          Y1, ZZZ2          Y2, ZZZ1
//   round 1   U1             U2               S1                S2
//   round 2   PP = P^2       ZZ1 ZZ2          ZZZ1 ZZZ2         RR = R^2
//   round 3   PPP            ZZ3              -                 Q = U1 PP
//   round 4   T2 = S1 PPP    -                ZZZ3              T1 = R (Q - X3)
//   result    Y3 = T1 - T2   ZZ3              ZZZ3              X3
// Returns false -- c untouched -- when the quad met equal x (p = +-q): the caller takes its general path.
template <class M>
__device__ __forceinline__ bool xyzz30_add_quadreg(F30<M>& c, const XYZZ<M>* q, const uint32_t* qx, bool neg, uint32_t r, uint32_t lane) {
    bool z;
    // lane 1: ZZ2, lane 2: X2, lane 3: ZZZ2, lane 0: Y2
    const int coord = r == 1u ? 2 : (r == 2u ? 0 : (r == 3u ? 3 : 1));
    const uint32_t* src = r == 2u ? qx : reinterpret_cast<const uint32_t*>(q) + 8 * coord;
    F30<M> L;
    {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        const uint4 a = s4[0], b = s4[1];
        const uint32_t t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        L = f30_unpack<M>(t);
    }
    (void)z;
    if (neg) L = f30_sel<M>(r == 0u, f30_sub<M, 4>(F30<M>{}, L), L);    // lane 0: 4p - Y2
    const F30<M> t = f30_quad<M, 0x63>(c);                                // quad_perm [3, 0, 2, 1]: lane 0 <- ZZZ1, 1 <- X1, 2 <- ZZ1, 3 <- Y1
    const bool first = (r & 1u) != 0u;                                    // lanes 1 and 3 hold the accumulator's coordinate as FIRST factor
    const F30<M> B = f30_sel<M>(first, L, t);                             // the ZZ / ZZZ factor: ZZ2, ZZ1, ZZZ2, ZZZ1 (lanes 1, 2, 3, 0)
    const F30<M> M1 = f30_mul<M>(t, L);                                   // U1, U2, S1, S2 (a product does not care which factor is whose)
    const bool edge = (r == 1u) || (r == 0u);                             // roles a and d
    const F30<M> rcv = f30_quad<M, 0x1B>(f30_sel<M>(edge, B, M1));        // quad_perm [3, 2, 1, 0]: a <-> b (lanes 1, 2), c <-> d (lanes 3, 0)
    const F30<M> D = f30_sub<M, 2>(f30_sel<M>(r == 1u, rcv, M1), f30_sel<M>(r == 1u, M1, rcv));   // lane 1: P = U2 - U1, lane 0: R = S2 - S1
    const F30<M> M2 = f30_mul<M>(f30_sel<M>(edge, D, B), f30_sel<M>(edge, D, rcv));               // PP, ZZ1 ZZ2, ZZZ1 ZZZ2, RR
    const bool same_x = r == 1u && f30_product_is_zero<M>(M2);
    if ((__ballot(same_x) >> (lane & 60u)) & 0xfull) return false;
    const F30<M> bPP = f30_quad<M, 0x55>(M2);
    const F30<M> bU1 = f30_quad<M, 0x55>(M1);
    const F30<M> M3 = f30_mul<M>(r == 1u ? D : (r == 0u ? bU1 : M2), r == 1u ? M2 : bPP);         // PPP, ZZ3, -, Q
    const F30<M> bPPP = f30_quad<M, 0x55>(M3);
    const F30<M> bS1 = f30_quad<M, 0xFF>(M1);
    const F30<M> E = f30_add2<M>(bPPP, M3);              // lane 0: PPP + 2Q
    const F30<M> X3 = f30_sub<M, 4>(M2, E);              // lane 0: RR - E
    const F30<M> Dq = f30_sub<M, 6>(M3, X3);             // lane 0: Q - X3
    const F30<M> M4 = f30_mul<M>(r == 1u ? bS1 : (r == 0u ? D : M2), r == 0u ? Dq : bPPP);        // T2, -, ZZZ3, T1
    const F30<M> bT1 = f30_quad<M, 0x00>(M4);
    const F30<M> Y3 = f30_sub<M, 2>(bT1, M4);            // lane 1: T1 - T2
    c = r == 0u ? X3 : (r == 1u ? Y3 : (r == 2u ? M3 : M4));
    return true;
}

// the accumulator as an ec.hip.h XYZZ in the 2^256 Montgomery form (canonical residues); infinity = all zero
template <class M>
__device__ __forceinline__ XYZZ<M> xyzz30_to_xyzz(const XYZZ30<M>& p) {
    XYZZ<M> r;
    if (p.inf) {
        r.x = fe_zero<M>(); r.y = fe_zero<M>(); r.zz = fe_zero<M>(); r.zzz = fe_zero<M>();
        return r;
    }
    const F30<M> c = f30_const<M>(M::R1);            // 2^256 mod p: x 2^270 * 2^256 / 2^270 = x 2^256
    r.x = f30_to_fe_canonical<M>(f30_mul<M>(p.x, c));
    r.y = f30_to_fe_canonical<M>(f30_mul<M>(p.y, c));
    r.zz = f30_to_fe_canonical<M>(f30_mul<M>(p.zz, c));
    r.zzz = f30_to_fe_canonical<M>(f30_mul<M>(p.zzz, c));
    return r;
}

}  // namespace porla
