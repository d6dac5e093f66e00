// Mixed addition of the bucket accumulation in the reduced-radix field form of fe30.cuh (30-bit limbs, radix 2^270).
// Same group law and exceptional cases as ec.cuh:xyzz_madd (the reference: gnark-crypto's g1JacExtended.addMixed behind
// G1Affine.MultiExp, porla/main.go:136) -- only the residues' representation differs, so the bucket sums are the same
// group elements and the MSM result stays bit-exact.
//
// Value bounds (multiples of p; a product's result is < p + 2^246, written "1"):
//   X1 <= 5, Y1 <= 3 (the accumulator, see X3 / Y3 below), ZZ1, ZZZ1 <= 1, X2, Y2 < 1
//   U2 = X2 ZZ1, S2 = Y2 ZZZ1                      <= 1
//   P  = U2 - X1 + 6p                              <= 7
//   R  = S2 - Y1 + 4p                              <= 5
//   PP = P^2, PPP = P PP, Q = X1 PP, RR = R^2      <= 1
//   E  = PPP + 2 Q                                 <= 3
//   X3 = RR - E + 4p                               <= 5
//   D  = Q - X3 + 6p                               <= 7
//   Y3 = R D - Y1 PPP + 2p                         <= 3
// Everything is below 8p < 2^257, so limb 8 stays below 2^18 as the products require.
#pragma once
#include "ec.cuh"
#include "fe30.cuh"

namespace porla {

template <class M>
struct XYZZ30 {
    F30<M> x, y, zz, zzz;
    bool inf;
};

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

// p += a (a affine in the 2^270 form, not infinity)
template <class M>
__device__ __forceinline__ void xyzz30_madd(XYZZ30<M>& p, const F30<M>& ax, const F30<M>& ay) {
    if (p.inf) {
        p.x = ax; p.y = ay;
        p.zz = f30_const<M>(M::R1_30); p.zzz = p.zz;
        p.inf = false;
        return;
    }
    F30<M> U2 = f30_mul<M>(ax, p.zz);
    F30<M> S2 = f30_mul<M>(ay, p.zzz);
    F30<M> Pp = f30_sub<M, 6>(U2, p.x);
    F30<M> Rr = f30_sub<M, 4>(S2, p.y);
    F30<M> PP = f30_sqr<M>(Pp);
    if (f30_product_is_zero<M>(PP)) {               // same x: the same point (double it) or its negative (infinity)
        F30<M> RR = f30_sqr<M>(Rr);
        if (f30_product_is_zero<M>(RR)) p = xyzz30_double_affine<M>(ax, ay);
        else p.inf = true;
        return;
    }
    F30<M> PPP = f30_mul<M>(Pp, PP);
    F30<M> Q = f30_mul<M>(p.x, PP);
    F30<M> RR = f30_sqr<M>(Rr);
    F30<M> E = f30_add2<M>(PPP, Q);
    F30<M> X3 = f30_sub<M, 4>(RR, E);
    F30<M> D = f30_sub<M, 6>(Q, X3);
    F30<M> T1 = f30_mul<M>(Rr, D);
    F30<M> T2 = f30_mul<M>(p.y, PPP);
    p.x = X3;
    p.y = f30_sub<M, 2>(T1, T2);
    p.zz = f30_mul<M>(p.zz, PP);
    p.zzz = f30_mul<M>(p.zzz, PPP);
}

// the accumulator as an ec.cuh XYZZ in the 2^256 Montgomery form (canonical residues); infinity = all zero
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
