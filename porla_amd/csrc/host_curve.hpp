// Host-side (CPU) tail of the engine: the few latency-bound group operations that stay on the host --
// projective -> affine normalisation, (un)marshalling, and the
// single-point operations of the plug-in (add_point / mult_point / neg_point, porla/main.go:195-230),
// which SURVEY.md s8(e) classifies as "replicas only" (one 64-byte operand: a PCIe round trip would cost
// more than the arithmetic).  It reuses the SAME limb code as the device (fe.hip.h / ec.hip.h are
// __host__ __device__), so the GPU parity tests exercise these formulas too.
#pragma once
#include "ec.hip.h"
#include <cstring>

namespace porla {

struct Bn254Fr {  // scalar field of BN254 (group order r), for KZG polynomial arithmetic on the host
    static constexpr uint32_t P[8]  = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                       0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV   = 0xefffffffu;
    static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                       0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr int SPARE_BITS = 2;
    static constexpr bool PSEUDO_MERSENNE = false;
    static constexpr uint32_t FOLD = 0;
};

// 32 big-endian bytes -> plain limbs (not reduced)
inline void h_load_be(uint32_t t[8], const uint8_t* b) {
    for (int i = 0; i < 8; i++) {
        const uint8_t* p = b + 4 * (7 - i);
        t[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
}
inline void h_store_be(uint8_t* b, const uint32_t t[8]) {
    for (int i = 0; i < 8; i++) {
        uint8_t* p = b + 4 * (7 - i);
        p[0] = (uint8_t)(t[i] >> 24); p[1] = (uint8_t)(t[i] >> 16); p[2] = (uint8_t)(t[i] >> 8); p[3] = (uint8_t)t[i];
    }
}

// SetBytes: big-endian -> reduced plain value -> Montgomery
template <class M>
inline Fe<M> h_fe_from_be(const uint8_t* b) {
    Fe<M> f;
    h_load_be(f.v, b);
    fe_reduce_plain<M>(f.v, 8);
    return fe_to_mont<M>(f);
}
// arbitrary-length big-endian byte string reduced mod P (fr.SetBytes on a 16-byte key, main.go:34,37)
template <class M>
inline Fe<M> h_fe_from_be_var(const uint8_t* b, size_t len) {
    Fe<M> acc = fe_zero<M>();
    Fe<M> k256 = fe_zero<M>();
    k256.v[0] = 256;
    k256 = fe_to_mont<M>(k256);
    for (size_t i = 0; i < len; i++) {
        Fe<M> d = fe_zero<M>();
        d.v[0] = b[i];
        d = fe_to_mont<M>(d);
        acc = fe_add<M>(fe_mul<M>(acc, k256), d);
    }
    return acc;
}
template <class M>
inline void h_fe_to_be(uint8_t* b, const Fe<M>& a) {
    Fe<M> p = fe_from_mont<M>(a);
    h_store_be(b, p.v);
}
// plain little-endian limbs of the regular value
template <class M>
inline void h_fe_to_plain(uint32_t out[8], const Fe<M>& a) {
    Fe<M> p = fe_from_mont<M>(a);
    for (int i = 0; i < 8; i++) out[i] = p.v[i];
}

template <class M>
inline Fe<M> h_fe_pow(const Fe<M>& a, const uint32_t e[8]) {
    Fe<M> acc = fe_one<M>();
    for (int i = 255; i >= 0; i--) {
        acc = fe_sqr<M>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul<M>(acc, a);
    }
    return acc;
}
template <class M>
inline Fe<M> h_fe_inv(const Fe<M>& a) {  // prime modulus: a^(p-2)
    uint32_t e[8];
    uint64_t br = 2;
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)M::P[i] - br;
        e[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
    return h_fe_pow<M>(a, e);
}

template <class M>
inline Affine<M> h_xyzz_to_affine(const XYZZ<M>& p) {
    Affine<M> r;
    if (xyzz_is_inf<M>(p)) { r.x = fe_zero<M>(); r.y = fe_zero<M>(); return r; }
    Fe<M> i = h_fe_inv<M>(fe_mul<M>(p.zz, p.zzz));
    r.x = fe_mul<M>(p.x, fe_mul<M>(i, p.zzz));
    r.y = fe_mul<M>(p.y, fe_mul<M>(i, p.zz));
    return r;
}

// G1Affine.Unmarshal, uncompressed form (flags 00): X, Y <- SetBytes; (0,0) = infinity
template <class M>
inline Affine<M> h_affine_from_bytes(const uint8_t* b) {
    Affine<M> a;
    a.x = h_fe_from_be<M>(b);
    a.y = h_fe_from_be<M>(b + 32);
    return a;
}
template <class M>
inline void h_affine_to_bytes(uint8_t* b, const Affine<M>& a) {
    if (aff_is_inf<M>(a)) { std::memset(b, 0, 64); return; }
    h_fe_to_be<M>(b, a.x);
    h_fe_to_be<M>(b + 32, a.y);
}

// k * a with k plain little-endian limbs (left-to-right double-and-add)
template <class M>
inline XYZZ<M> h_scalar_mul(const Affine<M>& a, const uint32_t k[8]) {
    XYZZ<M> acc = xyzz_inf<M>();
    int top = 255;
    while (top >= 0 && !((k[top >> 5] >> (top & 31)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        acc = xyzz_double<M>(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) xyzz_madd<M>(acc, a);
    }
    return acc;
}

// XYZZ -> Jacobian (X', Y', Z') with Z' = ZZ: X' = X*ZZ, Y' = Y*ZZZ; 96 bytes big-endian regular form
template <class M>
inline void h_xyzz_to_jac_bytes(uint8_t* b, const XYZZ<M>& p) {
    if (xyzz_is_inf<M>(p)) { std::memset(b, 0, 96); b[31] = 1; b[63] = 1; return; }
    h_fe_to_be<M>(b, fe_mul<M>(p.x, p.zz));
    h_fe_to_be<M>(b + 32, fe_mul<M>(p.y, p.zzz));
    h_fe_to_be<M>(b + 64, p.zz);
}
template <class M>
inline XYZZ<M> h_xyzz_from_jac_bytes(const uint8_t* b) {
    XYZZ<M> p;
    Fe<M> X = h_fe_from_be<M>(b), Y = h_fe_from_be<M>(b + 32), Z = h_fe_from_be<M>(b + 64);
    if (fe_is_zero<M>(Z)) return xyzz_inf<M>();
    p.x = X; p.y = Y; p.zz = fe_sqr<M>(Z); p.zzz = fe_mul<M>(p.zz, Z);
    return p;
}

}  // namespace porla
