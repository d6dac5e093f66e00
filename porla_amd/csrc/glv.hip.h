// GLV scalar decomposition for the two curves of the engine (y^2 = x^3 + b: phi(x, y) = (beta*x, y) = lambda*(x, y)).
//
// The reference's secp256k1 path splits every scalar the same way before its Pippenger pass
// (secp256k1_ecmult_endo_split, porla/Utils/secp256k1_lib/ecmult_impl.h:621-634 ->
// secp256k1_scalar_split_lambda, scalar_impl.h:123-156); gnark's BN254 MultiExp does not, the GLV form is used here for
// both because it halves the number of windows: half the buckets to reduce, half the doublings in the final fold, and for
// 256-bit secp256k1 scalars it removes the carry-only 17th window.  Only the group element sum (s_i mod n) * P_i matters
// for parity, and k = k1 + lambda*k2 (mod n) holds by construction for ANY rounding of c1, c2 below.
//
// Constants and the bit-for-bit Python model of glv_split: tools/gen_glv.py (self-checked on 2*10^5 scalars per curve).
#pragma once
#include "fe.hip.h"

namespace porla {

// Bn254: lambda = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
//   beta = 0x30644e72e131a0295e6dd9e7e0acccb0c28f069fbb966e3de4bd44e5607cfd48
//   basis (a1, b1) = (147946756881789319000765030803803410728, -9931322734385697763), (a2, b2) = (9931322734385697763, 147946756881789319010696353538189108491)
//   proven bound (|a1|+|a2|)/2, (|b1|+|b2|)/2 (+ rounding): 126 / 126 bits; worst seen over the self-check: 126 bits
struct GlvBn254 {
    static constexpr uint32_t BETA[8] = {0x607cfd48u, 0xe4bd44e5u, 0xbb966e3du, 0xc28f069fu, 0xe0acccb0u, 0x5e6dd9e7u, 0xe131a029u, 0x30644e72u};   // plain
    // beta in the field form the reduced-radix kernels compute in (fe30.hip.h): beta * 2^270 mod p
    static constexpr uint32_t BETA_30[8] = {0x553ba9feu, 0xf084d174u, 0x425467c7u, 0x743dbd42u, 0xac1030d9u, 0x64efb88du, 0x788d0b37u, 0x24f261b7u};
    static constexpr int SHIFT = 382;
    static constexpr uint32_t G1[8] = {0xf2d2e698u, 0x058ed210u, 0xf5792573u, 0x45275503u, 0xc03fd959u, 0x94e63f40u, 0x29dcf4b4u, 0x9333bc05u};     // round(2^SHIFT |b2| / n)
    static constexpr uint32_t G2[8] = {0x4bebee99u, 0xa3e9f4cbu, 0x1dce9bbcu, 0xdbae71c5u, 0xb1f82cf5u, 0xb64748cbu, 0x00000000u, 0x00000000u};     // round(2^SHIFT |b1| / n)
    static constexpr uint32_t A1[5] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u, 0x00000000u};   static constexpr bool A1_NEG = false;
    static constexpr uint32_t B1[5] = {0x94d213e3u, 0x89d32568u, 0x00000000u, 0x00000000u, 0x00000000u};   static constexpr bool B1_NEG = true;
    static constexpr uint32_t A2[5] = {0x94d213e3u, 0x89d32568u, 0x00000000u, 0x00000000u, 0x00000000u};   static constexpr bool A2_NEG = false;
    static constexpr uint32_t B2[5] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u, 0x00000000u};   static constexpr bool B2_NEG = false;
    static constexpr int BITS = 126;   // |k1|, |k2| < 2^BITS
};
// Secp256k1: lambda = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
//   beta = 0x7ae96a2b657c07106e64479eac3434e99cf0497512f58995c1396c28719501ee
//   basis (a1, b1) = (64502973549206556628585045361533709077, -303414439467246543595250775667605759171), (a2, b2) = (367917413016453100223835821029139468248, 64502973549206556628585045361533709077)
//   proven bound (|a1|+|a2|)/2, (|b1|+|b2|)/2 (+ rounding): 128 / 128 bits; worst seen over the self-check: 128 bits
struct GlvSecp256k1 {
    static constexpr uint32_t BETA[8] = {0x719501eeu, 0xc1396c28u, 0x12f58995u, 0x9cf04975u, 0xac3434e9u, 0x6e64479eu, 0x657c0710u, 0x7ae96a2bu};   // plain
    static constexpr uint32_t BETA_30[8] = {0x719501eeu, 0xc1396c28u, 0x12f58995u, 0x9cf04975u, 0xac3434e9u, 0x6e64479eu, 0x657c0710u, 0x7ae96a2bu};   // plain residues in that form too
    static constexpr int SHIFT = 384;
    static constexpr uint32_t G1[8] = {0x45dbb031u, 0xe893209au, 0x71e8ca7fu, 0x3daa8a14u, 0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u};     // round(2^SHIFT |b2| / n)
    static constexpr uint32_t G2[8] = {0x8ac47f71u, 0x1571b4aeu, 0x9df506c6u, 0x221208acu, 0x0abfe4c4u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u};     // round(2^SHIFT |b1| / n)
    static constexpr uint32_t A1[5] = {0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u, 0x00000000u};   static constexpr bool A1_NEG = false;
    static constexpr uint32_t B1[5] = {0x0abfe4c3u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u, 0x00000000u};   static constexpr bool B1_NEG = true;
    static constexpr uint32_t A2[5] = {0x9d44cfd8u, 0x57c1108du, 0xa8e2f3f6u, 0x14ca50f7u, 0x00000001u};   static constexpr bool A2_NEG = false;
    static constexpr uint32_t B2[5] = {0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u, 0x00000000u};   static constexpr bool B2_NEG = false;
    static constexpr int BITS = 128;   // |k1|, |k2| < 2^BITS
};

// (k * g + 2^(SHIFT-1)) >> SHIFT for 256-bit k, g: 128 bits of the rounded 512-bit product (the quotient is < 2^128)
template <int SHIFT>
PORLA_HD void glv_mul_shift(uint32_t out[4], const uint32_t k[8], const uint32_t g[8]) {
    uint32_t t[17];
#pragma unroll
    for (int i = 0; i < 17; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t x = (uint64_t)k[i] * g[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)x;
            carry = x >> 32;
        }
        t[i + 8] = (uint32_t)carry;
    }
    constexpr int RL = (SHIFT - 1) / 32, RB = (SHIFT - 1) % 32;   // rounding bit
    uint32_t carry = 0;
#pragma unroll
    for (int i = RL; i < 16; i++) {
        uint64_t y = (uint64_t)t[i] + (i == RL ? (1u << RB) : 0u) + carry;
        t[i] = (uint32_t)y;
        carry = (uint32_t)(y >> 32);
    }
    constexpr int L = SHIFT / 32, R = SHIFT % 32;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t lo = (L + i < 17) ? t[L + i] : 0u;
        const uint32_t hi = (L + i + 1 < 17) ? t[L + i + 1] : 0u;
        out[i] = R == 0 ? lo : ((lo >> R) | (hi << ((32 - R) & 31)));
    }
}

// acc (8 limbs, two's complement mod 2^256) +/-= c (4 limbs) * m (5 limbs)
template <bool SUB>
PORLA_HD void glv_muladd(uint32_t acc[8], const uint32_t c[4], const uint32_t m[5]) {
    uint32_t p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            if (i + j < 8) {
                uint64_t x = (uint64_t)c[i] * m[j] + p[i + j] + carry;
                p[i + j] = (uint32_t)x;
                carry = x >> 32;
            }
        }
        if (i + 5 < 8) p[i + 5] = (uint32_t)carry;
    }
    if (SUB) {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = sbb32(acc[i], p[i], br);
    } else {
        uint32_t cy = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t s = (uint64_t)acc[i] + p[i] + cy;
            acc[i] = (uint32_t)s;
            cy = (uint32_t)(s >> 32);
        }
    }
}

PORLA_HD bool glv_abs(uint32_t mag[4], const uint32_t v[8]) {
    const bool neg = (v[7] >> 31) != 0;
    uint32_t cy = neg ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t s = (uint64_t)(neg ? ~v[i] : v[i]) + cy;
        mag[i] = (uint32_t)s;
        cy = (uint32_t)(s >> 32);
    }
    return neg;
}

// k (reduced mod n, 8 limbs) -> |k1|, |k2| (4 limbs each, < 2^G::BITS) and their signs, k = k1 + lambda*k2 (mod n)
template <class G>
PORLA_HD void glv_split(const uint32_t k[8], uint32_t m1[4], bool& neg1, uint32_t m2[4], bool& neg2) {
    uint32_t g1[8], g2[8], a1[5], b1[5], a2[5], b2[5];
#pragma unroll
    for (int i = 0; i < 8; i++) { g1[i] = G::G1[i]; g2[i] = G::G2[i]; }
#pragma unroll
    for (int i = 0; i < 5; i++) { a1[i] = G::A1[i]; b1[i] = G::B1[i]; a2[i] = G::A2[i]; b2[i] = G::B2[i]; }
    uint32_t c1[4], c2[4];
    glv_mul_shift<G::SHIFT>(c1, k, g1);   // |c1|, c1 = S1 * |c1| with S1 = sign(b2)
    glv_mul_shift<G::SHIFT>(c2, k, g2);   // |c2|, c2 = S2 * |c2| with S2 = sign(-b1)
    constexpr bool S1_NEG = G::B2_NEG;
    constexpr bool S2_NEG = !G::B1_NEG;
    // k1 = k - c1*a1 - c2*a2 ;  k2 = -c1*b1 - c2*b2
    uint32_t v1[8], v2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { v1[i] = k[i]; v2[i] = 0; }
    glv_muladd<!(S1_NEG != G::A1_NEG)>(v1, c1, a1);   // subtract when c1*a1 is positive
    glv_muladd<!(S2_NEG != G::A2_NEG)>(v1, c2, a2);
    glv_muladd<!(S1_NEG != G::B1_NEG)>(v2, c1, b1);
    glv_muladd<!(S2_NEG != G::B2_NEG)>(v2, c2, b2);
    neg1 = glv_abs(m1, v1);
    neg2 = glv_abs(m2, v2);
}

}  // namespace porla
