// Host tail of the MSM in 4 x 64-bit limbs: the Horner fold of the tree's single-bit sums (W*c dependent doublings and additions)
// is a purely sequential chain, so it runs on one host core -- with 64x64->128 products it costs ~60 us instead of the
// ~270 us of the portable 8 x 32-bit code shared with the device.  Same Montgomery radix (R = 2^256) as fe.hip.h, so
// elements convert by repacking limbs.  The reference does this fold inside gnark's MultiExp / at the end of
// secp256k1_ecmult_pippenger_wnaf (porla/Utils/secp256k1_lib/ecmult_impl.h:544-564).
#pragma once
#include "ec.hip.h"
#include "glv.hip.h"
#include <memory>
#include <mutex>
#include <vector>

namespace porla {

template <class M>
struct Fp64 {
    typedef unsigned __int128 u128;
    uint64_t p[4];
    uint64_t inv;  // -p^-1 mod 2^64
    Fp64() {
        for (int i = 0; i < 4; i++) p[i] = ((uint64_t)M::P[2 * i + 1] << 32) | M::P[2 * i];
        uint64_t x = 1;  // Newton: x <- x * (2 - p0 * x), doubles the number of correct low bits
        for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
        inv = 0 - x;
        adx = false;
#if PORLA_FP64_ADX
        // the mulx / adcx / adox product below: needs those instructions and a modulus whose top bit is clear
        adx = !M::PSEUDO_MERSENNE && (p[3] >> 63) == 0 && host_has_adx();
        adx_pm = M::PSEUDO_MERSENNE && host_has_adx();
#endif
    }
    bool adx, adx_pm = false;
    struct E { uint64_t v[4]; };

    static E from(const Fe<M>& a) {
        E r;
        for (int i = 0; i < 4; i++) r.v[i] = ((uint64_t)a.v[2 * i + 1] << 32) | a.v[2 * i];
        return r;
    }
    static Fe<M> to(const E& a) {
        Fe<M> r;
        for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)a.v[i]; r.v[2 * i + 1] = (uint32_t)(a.v[i] >> 32); }
        return r;
    }
    static bool is_zero(const E& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }

    // t - p if t >= p (t given with an extra carry word)
    E cond_sub(const uint64_t t[4], uint64_t carry) const {
        unsigned long long s0, s1, s2, s3, br;
        s0 = __builtin_subcll(t[0], p[0], 0, &br);
        s1 = __builtin_subcll(t[1], p[1], br, &br);
        s2 = __builtin_subcll(t[2], p[2], br, &br);
        s3 = __builtin_subcll(t[3], p[3], br, &br);
        const bool ge = carry != 0 || br == 0;
        E r;
        r.v[0] = ge ? s0 : t[0]; r.v[1] = ge ? s1 : t[1]; r.v[2] = ge ? s2 : t[2]; r.v[3] = ge ? s3 : t[3];
        return r;
    }
    E add(const E& a, const E& b) const {
        unsigned long long c;
        uint64_t t[4];
        t[0] = __builtin_addcll(a.v[0], b.v[0], 0, &c);
        t[1] = __builtin_addcll(a.v[1], b.v[1], c, &c);
        t[2] = __builtin_addcll(a.v[2], b.v[2], c, &c);
        t[3] = __builtin_addcll(a.v[3], b.v[3], c, &c);
        return cond_sub(t, c);
    }
    E sub(const E& a, const E& b) const {
        unsigned long long br, c;
        uint64_t t[4];
        t[0] = __builtin_subcll(a.v[0], b.v[0], 0, &br);
        t[1] = __builtin_subcll(a.v[1], b.v[1], br, &br);
        t[2] = __builtin_subcll(a.v[2], b.v[2], br, &br);
        t[3] = __builtin_subcll(a.v[3], b.v[3], br, &br);
        const uint64_t mask = 0 - (uint64_t)br;                 // borrowed: add p back
        E r;
        r.v[0] = __builtin_addcll(t[0], p[0] & mask, 0, &c);
        r.v[1] = __builtin_addcll(t[1], p[1] & mask, c, &c);
        r.v[2] = __builtin_addcll(t[2], p[2] & mask, c, &c);
        r.v[3] = __builtin_addcll(t[3], p[3] & mask, c, &c);
        return r;
    }
#if PORLA_FP64_ADX
    // 256 x 256 -> 512-bit schoolbook product, one row per b[i] with two carry chains: T[i..i+4] += a * b[i]
#define PORLA_MUL_ROW(BI, T0, T1, T2, T3, T4)                                                   \
    "movq " BI "(%[b]), %%rdx\n\t"                                                              \
    "xorl %%eax, %%eax\n\t"                                                                     \
    "mulxq 0(%[a]), %[l], %[h]\n\t"  "adoxq %[l], " T0 "\n\t" "adcxq %[h], " T1 "\n\t"            \
    "mulxq 8(%[a]), %[l], %[h]\n\t"  "adoxq %[l], " T1 "\n\t" "adcxq %[h], " T2 "\n\t"            \
    "mulxq 16(%[a]), %[l], %[h]\n\t" "adoxq %[l], " T2 "\n\t" "adcxq %[h], " T3 "\n\t"            \
    "mulxq 24(%[a]), %[l], %[h]\n\t" "adoxq %[l], " T3 "\n\t" "adcxq %[h], " T4 "\n\t"            \
    "adoxq %%rax, " T4 "\n\t"
    __attribute__((target("bmi2,adx"))) static void mul512_adx(uint64_t t[8], const E& a, const E& b) {
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, l, h;
        asm(PORLA_MUL_ROW("0", "%[t0]", "%[t1]", "%[t2]", "%[t3]", "%[t4]")
            PORLA_MUL_ROW("8", "%[t1]", "%[t2]", "%[t3]", "%[t4]", "%[t5]")
            PORLA_MUL_ROW("16", "%[t2]", "%[t3]", "%[t4]", "%[t5]", "%[t6]")
            PORLA_MUL_ROW("24", "%[t3]", "%[t4]", "%[t5]", "%[t6]", "%[t7]")
            : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [t5] "+&r"(t5), [t6] "+&r"(t6),
              [t7] "+&r"(t7), [l] "=&r"(l), [h] "=&r"(h)
            : [a] "r"(a.v), [b] "r"(b.v), "m"(a), "m"(b)
            : "rax", "rdx", "cc");
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3; t[4] = t4; t[5] = t5; t[6] = t6; t[7] = t7;
    }
#undef PORLA_MUL_ROW
#endif
    // special-form product for p = 2^256 - 2^32 - FOLD on plain residues (fe_mul_pseudo_mersenne in fe.hip.h)
    E mul_pseudo_mersenne(const E& a, const E& b) const {
        const uint64_t c = ((uint64_t)1 << 32) + M::FOLD;
        uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#if PORLA_FP64_ADX
        if (adx_pm) mul512_adx(t, a, b);
        else
#endif
        for (int i = 0; i < 4; i++) {
            u128 carry = 0;
            for (int j = 0; j < 4; j++) { carry += (u128)a.v[i] * b.v[j] + t[i + j]; t[i + j] = (uint64_t)carry; carry >>= 64; }
            t[i + 4] = (uint64_t)carry;
        }
        uint64_t s[5];
        u128 carry = 0;
        for (int i = 0; i < 4; i++) { carry += (u128)t[4 + i] * c + t[i]; s[i] = (uint64_t)carry; carry >>= 64; }
        s[4] = (uint64_t)carry;                               // < 2^34
        u128 x = (u128)s[4] * c + s[0];
        uint64_t r[4];
        r[0] = (uint64_t)x; x >>= 64;
        for (int i = 1; i < 4; i++) { x += s[i]; r[i] = (uint64_t)x; x >>= 64; }
        if ((uint64_t)x) {                                    // 2^256 = c (mod p) once more; cannot carry again
            u128 y = (u128)r[0] + c;
            r[0] = (uint64_t)y; y >>= 64;
            for (int i = 1; i < 4; i++) { y += r[i]; r[i] = (uint64_t)y; y >>= 64; }
        }
        return cond_sub(r, 0);
    }
#if PORLA_FP64_ADX
    E mul_adx(const E& a, const E& b) const {          // fe.hip.h:mont_mul4_adx
        uint64_t t[4];
        mont_mul4_adx(t, a.v, b.v, p, inv);
        return cond_sub(t, 0);
    }
#endif
    // CIOS Montgomery product
    E mul(const E& a, const E& b) const {
        if (M::PSEUDO_MERSENNE) return mul_pseudo_mersenne(a, b);
#if PORLA_FP64_ADX
        if (adx) return mul_adx(a, b);     // 25 against 36 ns per dependent product, 18 against 28 ns with four products in flight (build container)
#endif
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * inv;
            c = (u128)m * p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) { c += (u128)m * p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        return cond_sub(t, t[4]);
    }

    E one() const { return from(fe_one<M>()); }
    // 1 / a in the field's own domain (Montgomery form in, Montgomery form out; plain for the special-form modulus): binary
    // extended Euclid on the stored integer -- the whole affine conversion takes 2.6 us with it against 13.8 us with Fermat's
    // a^(p-2) in the portable 8 x 32-bit code (build container), and it sits at the end of EVERY MSM / commitment call.  The stored integer
    // of a is a R, its plain inverse a^-1 R^-1; two products with R^2 bring that to a^-1 R (R^2 = 1 for the plain domain).
    E inverse(const E& a) const {
        if (is_zero(a)) return a;
        uint64_t u[4], v[4];
        E x1, x2;
        for (int i = 0; i < 4; i++) { u[i] = a.v[i]; v[i] = p[i]; x1.v[i] = 0; x2.v[i] = 0; }
        x1.v[0] = 1;
        auto is_one = [](const uint64_t* x) { return x[0] == 1 && (x[1] | x[2] | x[3]) == 0; };
        auto shr1 = [](uint64_t* x, uint64_t top) {
            x[0] = (x[0] >> 1) | (x[1] << 63); x[1] = (x[1] >> 1) | (x[2] << 63); x[2] = (x[2] >> 1) | (x[3] << 63);
            x[3] = (x[3] >> 1) | (top << 63);
        };
        auto halve = [&](E& x) {                       // x / 2 mod p
            uint64_t carry = 0;
            if (x.v[0] & 1) {
                u128 c = 0;
                for (int i = 0; i < 4; i++) { c += (u128)x.v[i] + p[i]; x.v[i] = (uint64_t)c; c >>= 64; }
                carry = (uint64_t)c;
            }
            shr1(x.v, carry);
        };
        auto ge = [](const uint64_t* x, const uint64_t* y) {
            for (int i = 3; i >= 0; i--) if (x[i] != y[i]) return x[i] > y[i];
            return true;
        };
        auto sub_in_place = [](uint64_t* x, const uint64_t* y) {
            u128 br = 0;
            for (int i = 0; i < 4; i++) { u128 d = (u128)x[i] - y[i] - (uint64_t)br; x[i] = (uint64_t)d; br = (d >> 64) & 1; }
        };
        while (!is_one(u) && !is_one(v)) {
            while (!(u[0] & 1)) { shr1(u, 0); halve(x1); }
            while (!(v[0] & 1)) { shr1(v, 0); halve(x2); }
            if (ge(u, v)) { sub_in_place(u, v); x1 = sub(x1, x2); }
            else { sub_in_place(v, u); x2 = sub(x2, x1); }
        }
        Fe<M> r2f;
        for (int i = 0; i < 8; i++) r2f.v[i] = M::R2[i];
        const E r2 = from(r2f);
        return mul(mul(is_one(u) ? x1 : x2, r2), r2);
    }
    // a^(p-2) (Fermat), kept as the cross-check of inverse()
    E inverse_fermat(const E& a) const {
        uint64_t e[4] = {p[0] - 2, p[1], p[2], p[3]};          // p is odd and p[0] >= 2: no borrow
        E acc = one();
        for (int l = 3; l >= 0; l--)
            for (int b = 63; b >= 0; b--) {
                acc = mul(acc, acc);
                if ((e[l] >> b) & 1) acc = mul(acc, a);
            }
        return acc;
    }

    struct Pt { E x, y, zz, zzz; };
    Pt from(const XYZZ<M>& q) const { Pt r; r.x = from(q.x); r.y = from(q.y); r.zz = from(q.zz); r.zzz = from(q.zzz); return r; }
    XYZZ<M> to(const Pt& q) const { XYZZ<M> r; r.x = to(q.x); r.y = to(q.y); r.zz = to(q.zz); r.zzz = to(q.zzz); return r; }
    Pt inf() const { Pt r; r.x = from(fe_one<M>()); r.y = r.x; r.zz = from(fe_zero<M>()); r.zzz = r.zz; return r; }

    // dbl-2008-s-1 (a = 0), same formulas as xyzz_double in ec.hip.h
    Pt dbl(const Pt& q) const {
        if (is_zero(q.zz) || is_zero(q.y)) return inf();
        Pt r;
        E U = add(q.y, q.y), V = mul(U, U), W = mul(U, V), S = mul(q.x, V), XX = mul(q.x, q.x);
        E Mm = add(add(XX, XX), XX);
        r.x = sub(sub(mul(Mm, Mm), S), S);
        r.y = sub(mul(Mm, sub(S, r.x)), mul(W, q.y));
        r.zz = mul(V, q.zz);
        r.zzz = mul(W, q.zzz);
        return r;
    }
    // add-2008-s, all exceptional cases as xyzz_add in ec.hip.h
    Pt padd(const Pt& a, const Pt& b) const {
        if (is_zero(b.zz)) return a;
        if (is_zero(a.zz)) return b;
        E U1 = mul(a.x, b.zz), U2 = mul(b.x, a.zz), S1 = mul(a.y, b.zzz), S2 = mul(b.y, a.zzz);
        E Pp = sub(U2, U1), Rr = sub(S2, S1);
        if (is_zero(Pp)) return is_zero(Rr) ? dbl(a) : inf();
        E PP = mul(Pp, Pp), PPP = mul(Pp, PP), Q = mul(U1, PP);
        Pt r;
        r.x = sub(sub(sub(mul(Rr, Rr), PPP), Q), Q);
        r.y = sub(mul(Rr, sub(Q, r.x)), mul(S1, PPP));
        r.zz = mul(mul(a.zz, b.zz), PP);
        r.zzz = mul(mul(a.zzz, b.zzz), PPP);
        return r;
    }
};

// one projective sum -> affine in 4 x 64-bit limbs (host_curve.hpp:h_xyzz_to_affine is the same on 8 x 32-bit limbs with Fermat's
// inversion: 13.8 us at the end of every call, 2.6 us here)
template <class M>
inline Affine<M> h_xyzz_to_affine64(const XYZZ<M>& q) {
    static const Fp64<M> F;
    typedef typename Fp64<M>::E E;
    Affine<M> r;
    if (xyzz_is_inf<M>(q)) { r.x = fe_zero<M>(); r.y = fe_zero<M>(); return r; }
    const E zz = Fp64<M>::from(q.zz), zzz = Fp64<M>::from(q.zzz);
    const E i = F.inverse(F.mul(zz, zzz));
    r.x = Fp64<M>::to(F.mul(Fp64<M>::from(q.x), F.mul(i, zzz)));      // X / ZZ
    r.y = Fp64<M>::to(F.mul(Fp64<M>::from(q.y), F.mul(i, zz)));       // Y / ZZZ
    return r;
}

// k * a on the host in 4 x 64-bit limbs, signed 4-bit windows (65 digits cover the carry): 256 doublings + <= 65 additions.
// The single-point operations of the plug-in (mult_point, compute_digest: porla/main.go:70-89, 205-213) -- the reference issues
// them one cgo call at a time from 8 threads; host_curve.hpp:h_scalar_mul is the same by double-and-add on 8 x 32-bit limbs.
template <class M>
inline XYZZ<M> h_scalar_mul64(const Affine<M>& a, const uint32_t k[8]) {
    static const Fp64<M> F;
    typedef typename Fp64<M>::Pt Pt;
    if (aff_is_inf<M>(a)) return xyzz_inf<M>();
    uint32_t nz = 0;
    for (int i = 0; i < 8; i++) nz |= k[i];
    if (!nz) return xyzz_inf<M>();
    Pt tbl[8];
    tbl[0].x = Fp64<M>::from(a.x); tbl[0].y = Fp64<M>::from(a.y); tbl[0].zz = F.one(); tbl[0].zzz = F.one();
    for (int i = 1; i < 8; i++) tbl[i] = F.padd(tbl[i - 1], tbl[0]);
    int8_t dig[65];
    uint32_t carry = 0;
    for (int i = 0; i < 64; i++) {
        const uint32_t d = ((k[i >> 3] >> ((i & 7) * 4)) & 15u) + carry;
        if (d > 8) { dig[i] = (int8_t)((int)d - 16); carry = 1; }
        else { dig[i] = (int8_t)d; carry = 0; }
    }
    dig[64] = (int8_t)carry;
    Pt acc = F.inf();
    typename Fp64<M>::E zero;
    for (int i = 0; i < 4; i++) zero.v[i] = 0;
    for (int i = 64; i >= 0; i--) {
        if (!Fp64<M>::is_zero(acc.zz)) for (int d = 0; d < 4; d++) acc = F.dbl(acc);
        const int d = dig[i];
        if (d > 0) acc = F.padd(acc, tbl[d - 1]);
        else if (d < 0) { Pt t = tbl[-d - 1]; t.y = F.sub(zero, t.y); acc = F.padd(acc, t); }
    }
    return F.to(acc);
}

// The same with the scalar split by the curve's endomorphism (glv.hip.h: k = k1 + k2 lambda, |k1|, |k2| < 2^128, phi(x, y) = (beta x, y)
// = lambda (x, y)): 33 windows of 4 doublings with two additions each instead of 65 with one -- half the doublings (mult_point
// 82 -> ~50 us).  k must be reduced modulo the group order (glv_split's precondition; mult_point and the MAC scaling reduce first).
template <class M, class G>
inline XYZZ<M> h_scalar_mul64_glv(const Affine<M>& a, const uint32_t k[8]) {
    static const Fp64<M> F;
    typedef typename Fp64<M>::Pt Pt;
    typedef typename Fp64<M>::E E;
    if (aff_is_inf<M>(a)) return xyzz_inf<M>();
    uint32_t nz = 0;
    for (int i = 0; i < 8; i++) nz |= k[i];
    if (!nz) return xyzz_inf<M>();
    static const E beta = [] {                       // beta in the field form of Fp64 (times R^2 / R; R = 1 for the special-form modulus)
        Fe<M> b, r2;
        for (int i = 0; i < 8; i++) { b.v[i] = G::BETA[i]; r2.v[i] = M::R2[i]; }
        return F.mul(Fp64<M>::from(b), Fp64<M>::from(r2));
    }();
    uint32_t m[2][4];
    bool ng[2];
    glv_split<G>(k, m[0], ng[0], m[1], ng[1]);
    Pt tbl[2][8];
    tbl[0][0].x = Fp64<M>::from(a.x); tbl[0][0].y = Fp64<M>::from(a.y); tbl[0][0].zz = F.one(); tbl[0][0].zzz = F.one();
    for (int i = 1; i < 8; i++) tbl[0][i] = F.padd(tbl[0][i - 1], tbl[0][0]);
    for (int i = 0; i < 8; i++) { tbl[1][i] = tbl[0][i]; tbl[1][i].x = F.mul(tbl[1][i].x, beta); }
    int8_t dig[2][33];
    for (int h = 0; h < 2; h++) {
        uint32_t carry = 0;
        for (int i = 0; i < 32; i++) {
            const uint32_t d = ((m[h][i >> 3] >> ((i & 7) * 4)) & 15u) + carry;
            if (d > 8) { dig[h][i] = (int8_t)((int)d - 16); carry = 1; }
            else { dig[h][i] = (int8_t)d; carry = 0; }
        }
        dig[h][32] = (int8_t)carry;
    }
    Pt acc = F.inf();
    E zero;
    for (int i = 0; i < 4; i++) zero.v[i] = 0;
    for (int i = 32; i >= 0; i--) {
        if (!Fp64<M>::is_zero(acc.zz)) for (int d = 0; d < 4; d++) acc = F.dbl(acc);
        for (int h = 0; h < 2; h++) {
            const int d = dig[h][i];
            if (!d) continue;
            Pt t = tbl[h][(d < 0 ? -d : d) - 1];
            if ((d < 0) != ng[h]) t.y = F.sub(zero, t.y);
            acc = F.padd(acc, t);
        }
    }
    return F.to(acc);
}

// n projective sums -> affine with ONE inversion (Montgomery's trick) in 4 x 64-bit limbs: the host tail of a batch of
// commitments (a row's own inversion costs ~25 us in the portable 8 x 32-bit code, which dominated a coalesced batch of 8)
template <class M>
inline void h_batch_xyzz_to_affine64(const XYZZ<M>* in, size_t n, Affine<M>* out) {
    static const Fp64<M> F;
    typedef typename Fp64<M>::E E;
    constexpr size_t MAXN = 64;
    E d[MAXN], pre[MAXN];
    for (size_t base = 0; base < n; base += MAXN) {
        const size_t m = n - base < MAXN ? n - base : MAXN;
        E acc = F.one();
        for (size_t i = 0; i < m; i++) {
            const XYZZ<M>& q = in[base + i];
            d[i] = xyzz_is_inf<M>(q) ? F.one() : F.mul(Fp64<M>::from(q.zz), Fp64<M>::from(q.zzz));
            pre[i] = acc;
            acc = F.mul(acc, d[i]);
        }
        E inv = F.inverse(acc);
        for (size_t i = m; i-- > 0;) {
            const XYZZ<M>& q = in[base + i];
            const E di = F.mul(inv, pre[i]);                     // 1 / (ZZ ZZZ) of entry i
            inv = F.mul(inv, d[i]);
            Affine<M>& r = out[base + i];
            if (xyzz_is_inf<M>(q)) { r.x = fe_zero<M>(); r.y = fe_zero<M>(); continue; }
            r.x = Fp64<M>::to(F.mul(Fp64<M>::from(q.x), F.mul(di, Fp64<M>::from(q.zzz))));     // X / ZZ
            r.y = Fp64<M>::to(F.mul(Fp64<M>::from(q.y), F.mul(di, Fp64<M>::from(q.zz))));      // Y / ZZZ
        }
    }
}

// k * B for a base point B that stays the same over many calls (compute_digest multiplies SRS.G1[0], compute_digest_complement
// the MAC hiding base: porla/main.go:87-88, 99-100): a table of j * 16^i * B (i < 65 signed 4-bit digits, j = 1 .. 8, affine)
// turns the multiplication into <= 65 mixed additions and NO doublings.  The table (33 KB) is rebuilt when the base changes.
template <class M>
struct HostFixedBase {
    typedef std::vector<Affine<M>> Table;          // [65][8]
    std::mutex mu;                                  // guards base / tab; the multiplication itself runs on a snapshot
    Affine<M> base;
    std::shared_ptr<const Table> tab;
    static bool same(const Affine<M>& a, const Affine<M>& b) {
        for (int i = 0; i < 8; i++) if (a.x.v[i] != b.x.v[i] || a.y.v[i] != b.y.v[i]) return false;
        return true;
    }
    static std::shared_ptr<const Table> build(const Affine<M>& b) {
        static const Fp64<M> F;
        typedef typename Fp64<M>::Pt Pt;
        std::vector<XYZZ<M>> proj(65 * 8);
        Pt P;
        P.x = Fp64<M>::from(b.x); P.y = Fp64<M>::from(b.y); P.zz = F.one(); P.zzz = F.one();
        for (int i = 0; i < 65; i++) {
            Pt acc = P;
            for (int j = 0; j < 8; j++) {
                proj[(size_t)i * 8 + j] = F.to(acc);
                if (j < 7) acc = F.padd(acc, P);
            }
            for (int d = 0; d < 4; d++) P = F.dbl(P);          // 16^(i+1) B
        }
        auto t = std::make_shared<Table>(65 * 8);
        h_batch_xyzz_to_affine64<M>(proj.data(), proj.size(), t->data());
        return t;
    }
    XYZZ<M> mul(const Affine<M>& b, const uint32_t k[8]) {
        static const Fp64<M> F;
        typedef typename Fp64<M>::Pt Pt;
        typedef typename Fp64<M>::E E;
        if (aff_is_inf<M>(b)) return xyzz_inf<M>();
        std::shared_ptr<const Table> t;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!tab || !same(base, b)) { tab = build(b); base = b; }
            t = tab;
        }
        Pt acc = F.inf();
        E zero;
        for (int i = 0; i < 4; i++) zero.v[i] = 0;
        uint32_t carry = 0;
        for (int i = 0; i < 65; i++) {
            uint32_t d = carry;
            if (i < 64) d += (k[i >> 3] >> ((i & 7) * 4)) & 15u;
            int dg;
            if (d > 8) { dg = (int)d - 16; carry = 1; } else { dg = (int)d; carry = 0; }
            if (dg == 0) continue;
            const Affine<M>& e = (*t)[(size_t)i * 8 + (size_t)((dg < 0 ? -dg : dg) - 1)];
            if (aff_is_inf<M>(e)) continue;
            Pt q;
            q.x = Fp64<M>::from(e.x); q.y = Fp64<M>::from(e.y); q.zz = F.one(); q.zzz = F.one();
            if (dg < 0) q.y = F.sub(zero, q.y);
            acc = F.padd(acc, q);
        }
        return F.to(acc);
    }
};

// Tree form of the bucket reduction (msm.hip.h, k_tree_level): window w arrives as fin[w][0] = S (sum of its buckets) and
// fin[w][1 + k] = M_k (sum of the buckets whose index has bit k set), k < c - 1, and is worth S + sum_k 2^k M_k.
// total = sum_w 2^(c*w) * that: one Horner pass over single bits -- the same W*c doublings, W*c additions.
template <class M>
inline XYZZ<M> h_fold_tree64(const XYZZ<M>* fin, int W, int c) {
    static const Fp64<M> F;
    typename Fp64<M>::Pt acc = F.inf();
    for (int w = W - 1; w >= 0; w--) {
        const XYZZ<M>* f = fin + (size_t)w * c;
        if (!Fp64<M>::is_zero(acc.zz)) acc = F.dbl(acc);          // bit c-1 of the bucket index does not exist
        for (int k = c - 2; k >= 0; k--) {
            if (!Fp64<M>::is_zero(acc.zz)) acc = F.dbl(acc);
            acc = F.padd(acc, F.from(f[1 + k]));
        }
        acc = F.padd(acc, F.from(f[0]));
    }
    return F.to(acc);
}

}  // namespace porla
