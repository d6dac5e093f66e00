// 256-bit prime-field arithmetic for gfx950 (CDNA4), 8 x 32-bit limbs, Montgomery form (R = 2^256).
//
// Replaces, on the device, the field layer the reference reaches through its third-party providers:
//   gnark-crypto v0.6.0 ecc/bn254/fp (call sites porla/main.go:130,136,200,212) and
//   libsecp256k1 field_5x52 (porla/Utils/secp256k1_lib/field_5x52_impl.h:432 fe_mul, :449 fe_sqr).
//
// Design notes (MI355X):
//  * There is no dense contraction here, so no MFMA: a field product is 64 + 64 v_mad_u64_u32
//    (32x32+64 -> 64, quarter-rate VALU) plus carry bookkeeping; everything lives in VGPRs.
//  * gfx950 needs two wait states between a VALU write of VCC/SGPR and a VALU read of it, so long
//    v_addc chains stall a lone wave.  The product-scanning (Comba) form below keeps one 96-bit column
//    accumulator and lets hipcc interleave the carry updates with the next multiply.
//  * All loops are fully unrolled with compile-time indices: limbs never leave registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// force-inline in device code only: the host tail is latency-tolerant and compiles much faster without it
#if defined(__HIP_DEVICE_COMPILE__)
#define PORLA_HD __host__ __device__ __forceinline__
#else
#define PORLA_HD __host__ __device__ inline
#endif

namespace porla {

// ---------------------------------------------------------------- field parameter packs
// BN254 base field (alt_bn128 p).  Constants cross-checked in tests against values derived
// from the modulus alone by the oracle (oracle/mont256.h mod256_init).
struct Bn254Fp {
    static constexpr uint32_t P[8]  = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                       0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV   = 0xe4866389u;  // -p^-1 mod 2^32
    static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                       0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};  // R mod p
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};  // R^2 mod p
    static constexpr int SPARE_BITS = 2;            // p < 2^254
};

// secp256k1 base field p = 2^256 - 2^32 - 977 (field_5x52.h:13-15 of the vendored tree).
struct Secp256k1Fp {
    static constexpr uint32_t P[8]  = {0xfffffc2fu, 0xfffffffeu, 0xffffffffu, 0xffffffffu,
                                       0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    static constexpr uint32_t INV   = 0xd2253531u;  // -p^-1 mod 2^32
    static constexpr uint32_t R1[8] = {0x000003d1u, 0x00000001u, 0, 0, 0, 0, 0, 0};          // 2^256 mod p
    static constexpr uint32_t R2[8] = {0x000e90a1u, 0x000007a2u, 0x00000001u, 0, 0, 0, 0, 0}; // (2^32+977)^2
    static constexpr int SPARE_BITS = 0;
};

// ---------------------------------------------------------------- element type
template <class M>
struct Fe {
    uint32_t v[8];
};

template <class M>
PORLA_HD bool fe_is_zero(const Fe<M>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
template <class M>
PORLA_HD bool fe_eq(const Fe<M>& a, const Fe<M>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
template <class M>
PORLA_HD Fe<M> fe_zero() {
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
template <class M>
PORLA_HD Fe<M> fe_one() {  // Montgomery one
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = M::R1[i];
    return r;
}

// t - P with borrow out; returns borrow (1 if t < P)
template <class M>
PORLA_HD uint32_t sub_p(uint32_t s[8], const uint32_t t[8]) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)t[i] - M::P[i] - br;
        s[i] = (uint32_t)d;
        br = (uint32_t)(d >> 63);
    }
    return br;
}

template <class M>
PORLA_HD Fe<M> fe_add(const Fe<M>& a, const Fe<M>& b) {
    uint32_t t[8], s[8];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)a.v[i] + b.v[i] + c;
        t[i] = (uint32_t)x;
        c = (uint32_t)(x >> 32);
    }
    uint32_t br = sub_p<M>(s, t);
    bool ge = (M::SPARE_BITS > 0) ? (br == 0) : (c != 0 || br == 0);
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? s[i] : t[i];
    return r;
}

template <class M>
PORLA_HD Fe<M> fe_sub(const Fe<M>& a, const Fe<M>& b) {
    uint32_t t[8];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.v[i] - b.v[i] - br;
        t[i] = (uint32_t)d;
        br = (uint32_t)(d >> 63);
    }
    uint32_t mask = 0u - br;  // add P back when the subtraction borrowed
    uint32_t c = 0;
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)t[i] + (M::P[i] & mask) + c;
        r.v[i] = (uint32_t)x;
        c = (uint32_t)(x >> 32);
    }
    return r;
}

template <class M>
PORLA_HD Fe<M> fe_neg(const Fe<M>& a) {
    Fe<M> z = fe_zero<M>();
    return fe_sub<M>(z, a);  // 0 - 0 = 0, else P - a
}

template <class M>
PORLA_HD Fe<M> fe_dbl(const Fe<M>& a) { return fe_add<M>(a, a); }

// 96-bit column accumulator step: (acc2:acc) += a*b
PORLA_HD void mac96(uint64_t& acc, uint32_t& acc2, uint32_t a, uint32_t b) {
    uint64_t t = acc + (uint64_t)a * b;
    acc2 += (t < acc) ? 1u : 0u;
    acc = t;
}

// Montgomery product a*b*R^-1 mod P, product-scanning with interleaved reduction (portable form: host pass,
// and the device pass when PORLA_NO_ASM_MUL is defined).
template <class M>
PORLA_HD Fe<M> fe_mul_generic(const Fe<M>& a, const Fe<M>& b) {
    uint32_t m[8], t[8], s[8];
    uint64_t acc = 0;
    uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j >= 0 && j < 8) mac96(acc, acc2, a.v[i], b.v[j]);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j >= 1 && j < 8 && i < k) mac96(acc, acc2, m[i], M::P[j]);
        }
        if (k < 8) {
            m[k] = (uint32_t)acc * M::INV;
            mac96(acc, acc2, m[k], M::P[0]);
        } else {
            t[k - 8] = (uint32_t)acc;
        }
        acc = (acc >> 32) | ((uint64_t)acc2 << 32);
        acc2 = 0;
    }
    uint32_t br = sub_p<M>(s, t);
    bool ge = (acc != 0) || (br == 0);
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? s[i] : t[i];
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(PORLA_NO_ASM_MUL)
#include "fe_mul_gfx950.inc"
#endif

template <class M>
PORLA_HD Fe<M> fe_mul(const Fe<M>& a, const Fe<M>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PORLA_NO_ASM_MUL)
    return fe_mul_gfx950<M>(a, b);
#else
    return fe_mul_generic<M>(a, b);
#endif
}

// (a dedicated square would save 28 of the 64 a_i*a_j products but needs a 96-bit shift and add per column to double the
// triangular sum -- measured by instruction count it is no faster than the general product on gfx950, so there is none)
template <class M>
PORLA_HD Fe<M> fe_sqr(const Fe<M>& a) { return fe_mul<M>(a, a); }

// Out-of-line product: one shared ~2.7 KB body per field instead of an inlined copy per use.  The cold reduction
// kernels chain dozens of group operations of 9-14 products each; fully inlined they exceed the instruction cache
// (64 KB per CU pair), called they fit in a few KB.
template <class M>
__host__ __device__
#if defined(__HIP_DEVICE_COMPILE__)
__noinline__
#else
inline
#endif
Fe<M> fe_mul_call(Fe<M> a, Fe<M> b) { return fe_mul<M>(a, b); }
template <class M>
__host__ __device__
#if defined(__HIP_DEVICE_COMPILE__)
__noinline__
#else
inline
#endif
Fe<M> fe_sqr_call(Fe<M> a) { return fe_sqr<M>(a); }
// CALL = true -> out-of-line product
template <class M, bool CALL>
PORLA_HD Fe<M> fmul(const Fe<M>& a, const Fe<M>& b) {
    if (CALL) return fe_mul_call<M>(a, b);
    return fe_mul<M>(a, b);
}
template <class M, bool CALL>
PORLA_HD Fe<M> fsqr(const Fe<M>& a) {
    if (CALL) return fe_sqr_call<M>(a);
    return fe_sqr<M>(a);
}

template <class M>
PORLA_HD Fe<M> fe_to_mont(const Fe<M>& a) {
    Fe<M> r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = M::R2[i];
    return fe_mul<M>(a, r2);
}
template <class M>
PORLA_HD Fe<M> fe_from_mont(const Fe<M>& a) {
    Fe<M> one = fe_zero<M>();
    one.v[0] = 1;
    return fe_mul<M>(a, one);
}

// 32 big-endian bytes (as four 64-bit... two uint4 loads) -> little-endian limbs, NOT reduced.
PORLA_HD uint32_t bswap32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bswap32(x);
#else
    return __builtin_bswap32(x);
#endif
}

// plain (non-Montgomery) value < 2^256 reduced into [0, P): at most `maxq` conditional subtractions
template <class M>
PORLA_HD void fe_reduce_plain(uint32_t t[8], int maxq) {
    for (int q = 0; q < maxq; q++) {
        uint32_t s[8];
        uint32_t br = sub_p<M>(s, t);
        if (br) break;
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = s[i];
    }
}

}  // namespace porla
