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
#include "fe.cuh"

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
template <class M>
PORLA_HD XYZZ<M> xyzz_double_affine(const Affine<M>& a) {
    XYZZ<M> r;
    if (fe_is_zero<M>(a.y)) return xyzz_inf<M>();  // order-2 point: cannot occur on prime-order curves
    Fe<M> U = fe_dbl<M>(a.y);
    Fe<M> V = fe_sqr<M>(U);
    Fe<M> W = fe_mul<M>(U, V);
    Fe<M> S = fe_mul<M>(a.x, V);
    Fe<M> XX = fe_sqr<M>(a.x);
    Fe<M> Mm = fe_add<M>(fe_dbl<M>(XX), XX);
    r.x = fe_sub<M>(fe_sub<M>(fe_sqr<M>(Mm), S), S);
    r.y = fe_sub<M>(fe_mul<M>(Mm, fe_sub<M>(S, r.x)), fe_mul<M>(W, a.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2 * p   (dbl-2008-s-1, a = 0)
template <class M>
PORLA_HD XYZZ<M> xyzz_double(const XYZZ<M>& p) {
    if (xyzz_is_inf<M>(p) || fe_is_zero<M>(p.y)) return xyzz_inf<M>();
    XYZZ<M> r;
    Fe<M> U = fe_dbl<M>(p.y);
    Fe<M> V = fe_sqr<M>(U);
    Fe<M> W = fe_mul<M>(U, V);
    Fe<M> S = fe_mul<M>(p.x, V);
    Fe<M> XX = fe_sqr<M>(p.x);
    Fe<M> Mm = fe_add<M>(fe_dbl<M>(XX), XX);
    r.x = fe_sub<M>(fe_sub<M>(fe_sqr<M>(Mm), S), S);
    r.y = fe_sub<M>(fe_mul<M>(Mm, fe_sub<M>(S, r.x)), fe_mul<M>(W, p.y));
    r.zz = fe_mul<M>(V, p.zz);
    r.zzz = fe_mul<M>(W, p.zzz);
    return r;
}

// p += a   (madd-2008-s), all exceptional cases handled
template <class M>
PORLA_HD void xyzz_madd(XYZZ<M>& p, const Affine<M>& a) {
    if (aff_is_inf<M>(a)) return;
    if (xyzz_is_inf<M>(p)) {
        p.x = a.x; p.y = a.y; p.zz = fe_one<M>(); p.zzz = fe_one<M>();
        return;
    }
    Fe<M> U2 = fe_mul<M>(a.x, p.zz);
    Fe<M> S2 = fe_mul<M>(a.y, p.zzz);
    Fe<M> Pp = fe_sub<M>(U2, p.x);
    Fe<M> Rr = fe_sub<M>(S2, p.y);
    if (fe_is_zero<M>(Pp)) {
        if (fe_is_zero<M>(Rr)) p = xyzz_double_affine<M>(a);
        else p = xyzz_inf<M>();
        return;
    }
    Fe<M> PP = fe_sqr<M>(Pp);
    Fe<M> PPP = fe_mul<M>(Pp, PP);
    Fe<M> Q = fe_mul<M>(p.x, PP);
    Fe<M> X3 = fe_sub<M>(fe_sub<M>(fe_sub<M>(fe_sqr<M>(Rr), PPP), Q), Q);
    Fe<M> Y3 = fe_sub<M>(fe_mul<M>(Rr, fe_sub<M>(Q, X3)), fe_mul<M>(p.y, PPP));
    p.x = X3;
    p.y = Y3;
    p.zz = fe_mul<M>(p.zz, PP);
    p.zzz = fe_mul<M>(p.zzz, PPP);
}

// p += q   (add-2008-s), all exceptional cases handled
template <class M>
PORLA_HD void xyzz_add(XYZZ<M>& p, const XYZZ<M>& q) {
    if (xyzz_is_inf<M>(q)) return;
    if (xyzz_is_inf<M>(p)) { p = q; return; }
    Fe<M> U1 = fe_mul<M>(p.x, q.zz);
    Fe<M> U2 = fe_mul<M>(q.x, p.zz);
    Fe<M> S1 = fe_mul<M>(p.y, q.zzz);
    Fe<M> S2 = fe_mul<M>(q.y, p.zzz);
    Fe<M> Pp = fe_sub<M>(U2, U1);
    Fe<M> Rr = fe_sub<M>(S2, S1);
    if (fe_is_zero<M>(Pp)) {
        if (fe_is_zero<M>(Rr)) p = xyzz_double<M>(p);
        else p = xyzz_inf<M>();
        return;
    }
    Fe<M> PP = fe_sqr<M>(Pp);
    Fe<M> PPP = fe_mul<M>(Pp, PP);
    Fe<M> Q = fe_mul<M>(U1, PP);
    Fe<M> X3 = fe_sub<M>(fe_sub<M>(fe_sub<M>(fe_sqr<M>(Rr), PPP), Q), Q);
    Fe<M> Y3 = fe_sub<M>(fe_mul<M>(Rr, fe_sub<M>(Q, X3)), fe_mul<M>(S1, PPP));
    p.x = X3;
    p.y = Y3;
    p.zz = fe_mul<M>(fe_mul<M>(p.zz, q.zz), PP);
    p.zzz = fe_mul<M>(fe_mul<M>(p.zzz, q.zzz), PPP);
}

// Out-of-line copies for the cold reduction kernels (keeps their code size and compile time bounded;
// the hot bucket accumulation inlines everything).
#if defined(__HIP_DEVICE_COMPILE__)
template <class M>
__device__ __noinline__ void xyzz_add_cold(XYZZ<M>* p, const XYZZ<M>* q) {
    XYZZ<M> a = *p;
    xyzz_add<M>(a, *q);
    *p = a;
}
template <class M>
__device__ __noinline__ void xyzz_double_cold(XYZZ<M>* p) {
    XYZZ<M> a = xyzz_double<M>(*p);
    *p = a;
}
#else
template <class M>
__host__ __device__ inline void xyzz_add_cold(XYZZ<M>* p, const XYZZ<M>* q) { xyzz_add<M>(*p, *q); }
template <class M>
__host__ __device__ inline void xyzz_double_cold(XYZZ<M>* p) { *p = xyzz_double<M>(*p); }
#endif

template <class M>
PORLA_HD Affine<M> aff_neg_if(const Affine<M>& a, bool neg) {
    Affine<M> r;
    r.x = a.x;
    Fe<M> ny = fe_neg<M>(a.y);
#pragma unroll
    for (int i = 0; i < 8; i++) r.y.v[i] = neg ? ny.v[i] : a.y.v[i];
    return r;
}

}  // namespace porla
