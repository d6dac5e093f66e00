// Curve arithmetic y^2 = x^3 + b (a = 0) for the MSM buckets, extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0).
//
// The reference gets this layer from gnark-crypto's g1JacExtended (bucket type of G1Affine.MultiExp,
// porla/main.go:136) and from libsecp256k1's gej_add_ge_var / gej_add_var / gej_double
// (porla/Utils/secp256k1_lib/group_impl.h:389-436, :336-387, :274-306).  Only the group element
// matters for parity, so the formulas are chosen for the GPU: the mixed add costs 8M + 2S and
// needs no inversion; every exceptional case (empty bucket, P + P, P + (-P), infinity operand)
// is handled, because real audit inputs repeat points (SURVEY.md s7 "hard parts" ii).
#pragma once
#include "fe.hip.h"

namespace porla {

template <class M>
struct Affine {  // Montgomery-form coordinates; (0,0) encodes the point at infinity
    Fe<M> x, y;
};

template <class M>
struct XYZZ {
    Fe<M> x, y, zz, zzz;
};

template <class M>
PORLA_HD bool aff_is_inf(const Affine<M>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.x.v[i] | a.y.v[i];
    return o == 0;
}
template <class M>
PORLA_HD bool xyzz_is_inf(const XYZZ<M>& p) { return fe_is_zero<M>(p.zz); }

template <class M>
PORLA_HD XYZZ<M> xyzz_inf() {
    XYZZ<M> r;
    r.x = fe_one<M>(); r.y = fe_one<M>(); r.zz = fe_zero<M>(); r.zzz = fe_zero<M>();
    return r;
}
template <class M>
PORLA_HD XYZZ<M> xyzz_from_affine(const Affine<M>& a) {
    if (aff_is_inf<M>(a)) return xyzz_inf<M>();
    XYZZ<M> r;
    r.x = a.x; r.y = a.y; r.zz = fe_one<M>(); r.zzz = fe_one<M>();
    return r;
}

// 2 * (affine a), a != infinity   (mdbl-2008-s-1, a = 0)
template <class M, bool CALL = false>
PORLA_HD XYZZ<M> xyzz_double_affine(const Affine<M>& a) {
    XYZZ<M> r;
    if (fe_is_zero<M>(a.y)) return xyzz_inf<M>();  // order-2 point: cannot occur on prime-order curves
    Fe<M> U = fe_dbl<M>(a.y);
    Fe<M> V = fsqr<M, CALL>(U);
    Fe<M> W = fmul<M, CALL>(U, V);
    Fe<M> S = fmul<M, CALL>(a.x, V);
    Fe<M> XX = fsqr<M, CALL>(a.x);
    Fe<M> Mm = fe_add<M>(fe_dbl<M>(XX), XX);
    r.x = fe_sub<M>(fe_sub<M>(fsqr<M, CALL>(Mm), S), S);
    r.y = fe_sub<M>(fmul<M, CALL>(Mm, fe_sub<M>(S, r.x)), fmul<M, CALL>(W, a.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2 * p   (dbl-2008-s-1, a = 0)
template <class M, bool CALL = false>
PORLA_HD XYZZ<M> xyzz_double(const XYZZ<M>& p) {
    if (xyzz_is_inf<M>(p) || fe_is_zero<M>(p.y)) return xyzz_inf<M>();
    XYZZ<M> r;
    Fe<M> U = fe_dbl<M>(p.y);
    Fe<M> V = fsqr<M, CALL>(U);
    Fe<M> W = fmul<M, CALL>(U, V);
    Fe<M> S = fmul<M, CALL>(p.x, V);
    Fe<M> XX = fsqr<M, CALL>(p.x);
    Fe<M> Mm = fe_add<M>(fe_dbl<M>(XX), XX);
    r.x = fe_sub<M>(fe_sub<M>(fsqr<M, CALL>(Mm), S), S);
    r.y = fe_sub<M>(fmul<M, CALL>(Mm, fe_sub<M>(S, r.x)), fmul<M, CALL>(W, p.y));
    r.zz = fmul<M, CALL>(V, p.zz);
    r.zzz = fmul<M, CALL>(W, p.zzz);
    return r;
}

// p += a   (madd-2008-s), all exceptional cases handled
template <class M, bool CALL = false>
PORLA_HD void xyzz_madd(XYZZ<M>& p, const Affine<M>& a) {
    if (aff_is_inf<M>(a)) return;
    if (xyzz_is_inf<M>(p)) {
        p.x = a.x; p.y = a.y; p.zz = fe_one<M>(); p.zzz = fe_one<M>();
        return;
    }
    Fe<M> U2 = fmul<M, CALL>(a.x, p.zz);
    Fe<M> S2 = fmul<M, CALL>(a.y, p.zzz);
    Fe<M> Pp = fe_sub<M>(U2, p.x);
    Fe<M> Rr = fe_sub<M>(S2, p.y);
    if (fe_is_zero<M>(Pp)) {
        if (fe_is_zero<M>(Rr)) p = xyzz_double_affine<M, CALL>(a);
        else p = xyzz_inf<M>();
        return;
    }
    Fe<M> PP = fsqr<M, CALL>(Pp);
    Fe<M> PPP = fmul<M, CALL>(Pp, PP);
    Fe<M> Q = fmul<M, CALL>(p.x, PP);
    Fe<M> X3 = fe_sub<M>(fe_sub<M>(fe_sub<M>(fsqr<M, CALL>(Rr), PPP), Q), Q);
    Fe<M> Y3 = fe_sub<M>(fmul<M, CALL>(Rr, fe_sub<M>(Q, X3)), fmul<M, CALL>(p.y, PPP));
    p.x = X3;
    p.y = Y3;
    p.zz = fmul<M, CALL>(p.zz, PP);
    p.zzz = fmul<M, CALL>(p.zzz, PPP);
}

// p += q   (add-2008-s), all exceptional cases handled
template <class M, bool CALL = false>
PORLA_HD void xyzz_add(XYZZ<M>& p, const XYZZ<M>& q) {
    if (xyzz_is_inf<M>(q)) return;
    if (xyzz_is_inf<M>(p)) { p = q; return; }
    Fe<M> U1 = fmul<M, CALL>(p.x, q.zz);
    Fe<M> U2 = fmul<M, CALL>(q.x, p.zz);
    Fe<M> S1 = fmul<M, CALL>(p.y, q.zzz);
    Fe<M> S2 = fmul<M, CALL>(q.y, p.zzz);
    Fe<M> Pp = fe_sub<M>(U2, U1);
    Fe<M> Rr = fe_sub<M>(S2, S1);
    if (fe_is_zero<M>(Pp)) {
        if (fe_is_zero<M>(Rr)) p = xyzz_double<M, CALL>(p);
        else p = xyzz_inf<M>();
        return;
    }
    Fe<M> PP = fsqr<M, CALL>(Pp);
    Fe<M> PPP = fmul<M, CALL>(Pp, PP);
    Fe<M> Q = fmul<M, CALL>(U1, PP);
    Fe<M> X3 = fe_sub<M>(fe_sub<M>(fe_sub<M>(fsqr<M, CALL>(Rr), PPP), Q), Q);
    Fe<M> Y3 = fe_sub<M>(fmul<M, CALL>(Rr, fe_sub<M>(Q, X3)), fmul<M, CALL>(S1, PPP));
    p.x = X3;
    p.y = Y3;
    p.zz = fmul<M, CALL>(fmul<M, CALL>(p.zz, q.zz), PP);
    p.zzz = fmul<M, CALL>(fmul<M, CALL>(p.zzz, q.zzz), PPP);
}

// Cold-path forms for the reduction kernels: the group law is inlined but every field product is a call to one shared
// out-of-line body (fe_mul_call), which keeps those kernels inside the instruction cache.
template <class M>
PORLA_HD void xyzz_add_cold(XYZZ<M>* p, const XYZZ<M>* q) { xyzz_add<M, true>(*p, *q); }
template <class M>
PORLA_HD void xyzz_double_cold(XYZZ<M>* p) { *p = xyzz_double<M, true>(*p); }

template <class M>
PORLA_HD Affine<M> aff_neg_if(const Affine<M>& a, bool neg) {
    Affine<M> r;
    r.x = a.x;
    r.y = fe_neg_if<M>(a.y, neg);
    return r;
}

}  // namespace porla
