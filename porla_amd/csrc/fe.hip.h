// 256-bit prime-field arithmetic for gfx950 (CDNA4), 8 x 32-bit limbs, Montgomery form (R = 2^256).
//
// Replaces, on the device, the field layer the reference reaches through its third-party providers:
//   gnark-crypto v0.6.0 ecc/bn254/fp (call sites porla/main.go:130,136,200,212) and
//   libsecp256k1 field_5x52 (porla/Utils/secp256k1_lib/field_5x52_impl.h:432 fe_mul, :449 fe_sqr).
//
// Design notes (MI355X):
//  * There is no dense contraction here, so no MFMA: a Montgomery product is 64 + 64 v_mad_u64_u32
//    (32x32+64 -> 64; measured 5.7 cycles per wave instruction) plus carry bookkeeping; everything lives in VGPRs.
//  * gfx950 needs two wait states between a VALU write of VCC/SGPR and a VALU read of it, so long
//    v_addc chains stall a lone wave.  The product-scanning (Comba) form keeps one 96-bit column accumulator; the
//    device form is generated assembly (fe_mul_gfx950.inc, tools/gen_fe_mul_asm.py) with rotating SGPR carries and
//    per-modulus variants (bounded limbs, sparse p_icc, special-form secp256k1), the portable form below is what the
//    host pass runs and what tools/fe_check.hip compares the assembly against.
//  * All loops are fully unrolled with compile-time indices: limbs never leave registers.
#pragma once
#include <cstdlib>
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
    static constexpr bool PSEUDO_MERSENNE = false;
    static constexpr uint32_t FOLD = 0;
    // reduced-radix form (fe30.hip.h, Montgomery radix 2^270): 2^270 mod p, and 2^(256+270) mod p -- the factor that takes a
    // plain residue into the 2^270 form through fe_mul (which divides by 2^256)
    static constexpr uint32_t R1_30[8] = {0x5accccc9u, 0xf89a7d3au, 0x9a1dcc9du, 0x50fea7bdu,
                                          0xdf1160f4u, 0x9e7d8ca3u, 0xed6d3304u, 0x279be39au};
    static constexpr uint32_t R2_30[8] = {0x5b61e465u, 0x0c82ea1du, 0xf298bc82u, 0x87f2ef02u,
                                          0x4050e665u, 0x9938cdbeu, 0xdf213851u, 0x2c026cedu};
    // 2^540 mod p: a plain residue times this in the reduced-radix product (radix 2^270) is the residue's 2^270 form
    static constexpr uint32_t RR_30[8] = {0x2e53b794u, 0x242db528u, 0x301f5ed1u, 0xa3522573u,
                                          0x9d4e3aa6u, 0x93560daau, 0x51b66a12u, 0x0d15816du};
    // 2^782 mod p = 2^(512 + 270): the plain inverse of the INTEGER a residue's Fe form holds (A 2^256), times this in the
    // reduced-radix product, is A^-1 2^256 -- the inverse in the Fe form (inv30.hip.h:fe_inv_safegcd)
    static constexpr uint32_t INV_OUT_30[8] = {0x27118959u, 0x136c05cau, 0x42d79087u, 0x7eef37ccu,
                                               0x568e8d6du, 0x5c91832eu, 0x53347fadu, 0x13616943u};
};

// secp256k1 base field p = 2^256 - 2^32 - 977 (field_5x52.h:13-15 of the vendored tree).  Like the reference's own field
// code this uses the special form of p: elements are PLAIN residues (the "Montgomery" radix is 1, so fe_to_mont /
// fe_from_mont are the identity and R1 = R2 = 1) and a product is a 512-bit product folded twice with 2^256 = 2^32 + 977
// (mod p) -- 64 + 10 multiply-adds instead of the 128 of a Montgomery product.
struct Secp256k1Fp {
    static constexpr uint32_t P[8]  = {0xfffffc2fu, 0xfffffffeu, 0xffffffffu, 0xffffffffu,
                                       0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    static constexpr uint32_t INV   = 0xd2253531u;  // -p^-1 mod 2^32 (unused by the special-form product)
    static constexpr uint32_t R1[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    static constexpr uint32_t R2[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    static constexpr int SPARE_BITS = 0;
    static constexpr bool PSEUDO_MERSENNE = true;   // p = 2^256 - (2^32 + FOLD)
    static constexpr uint32_t FOLD = 977;
    // reduced-radix form (fe30.hip.h): plain residues there too, so both conversion factors are 1
    static constexpr uint32_t R1_30[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    static constexpr uint32_t R2_30[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    static constexpr uint32_t RR_30[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    static constexpr uint32_t INV_OUT_30[8] = {1, 0, 0, 0, 0, 0, 0, 0};
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

// One word of a borrow / carry chain.  The device pass goes through the compiler's carry builtins, which become v_sub_co / v_subb_co
// (v_add_co / v_addc_co): ONE instruction per word.  The 64-bit difference form `d = (uint64_t)a - b - br; br = d >> 63` compiles to
// five per word on gfx950 (sign-extended 64-bit adds and the moves that build their operands: 38 instructions for an 8-word
// subtraction against 15) -- that was a twentieth of the ICC encode's instruction stream (its finish step: profiles/README.md, round 5).
PORLA_HD uint32_t sbb32(uint32_t a, uint32_t b, uint32_t& br) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned out;
    const uint32_t d = __builtin_subc(a, b, br, &out);
    br = out;
    return d;
#else
    const uint64_t d = (uint64_t)a - b - br;
    br = (uint32_t)(d >> 63);
    return (uint32_t)d;
#endif
}
PORLA_HD uint32_t adc32(uint32_t a, uint32_t b, uint32_t& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned out;
    const uint32_t x = __builtin_addc(a, b, c, &out);
    c = out;
    return x;
#else
    const uint64_t x = (uint64_t)a + b + c;
    c = (uint32_t)(x >> 32);
    return (uint32_t)x;
#endif
}

// t - P with borrow out; returns borrow (1 if t < P)
template <class M>
PORLA_HD uint32_t sub_p(uint32_t s[8], const uint32_t t[8]) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = sbb32(t[i], M::P[i], br);
    return br;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// host pass: the limbs taken two at a time (little-endian host: the same bytes), carries through the compiler's builtins --
// the pairing of verify_proof is as many additions as products
template <class M>
inline void fe_host64_load(unsigned long long w[4], const uint32_t v[8]) {
    for (int i = 0; i < 4; i++) w[i] = ((unsigned long long)v[2 * i + 1] << 32) | v[2 * i];
}
template <class M>
inline Fe<M> fe_add_host64(const Fe<M>& a, const Fe<M>& b) {
    unsigned long long A[4], B[4], Pm[4], t[4], s[4], c, br;
    fe_host64_load<M>(A, a.v); fe_host64_load<M>(B, b.v); fe_host64_load<M>(Pm, M::P);
    t[0] = __builtin_addcll(A[0], B[0], 0, &c); t[1] = __builtin_addcll(A[1], B[1], c, &c);
    t[2] = __builtin_addcll(A[2], B[2], c, &c); t[3] = __builtin_addcll(A[3], B[3], c, &c);
    s[0] = __builtin_subcll(t[0], Pm[0], 0, &br); s[1] = __builtin_subcll(t[1], Pm[1], br, &br);
    s[2] = __builtin_subcll(t[2], Pm[2], br, &br); s[3] = __builtin_subcll(t[3], Pm[3], br, &br);
    const bool ge = c != 0 || br == 0;
    Fe<M> r;
    for (int i = 0; i < 4; i++) { const unsigned long long w = ge ? s[i] : t[i]; r.v[2 * i] = (uint32_t)w; r.v[2 * i + 1] = (uint32_t)(w >> 32); }
    return r;
}
template <class M>
inline Fe<M> fe_sub_host64(const Fe<M>& a, const Fe<M>& b) {
    unsigned long long A[4], B[4], Pm[4], t[4], c, br;
    fe_host64_load<M>(A, a.v); fe_host64_load<M>(B, b.v); fe_host64_load<M>(Pm, M::P);
    t[0] = __builtin_subcll(A[0], B[0], 0, &br); t[1] = __builtin_subcll(A[1], B[1], br, &br);
    t[2] = __builtin_subcll(A[2], B[2], br, &br); t[3] = __builtin_subcll(A[3], B[3], br, &br);
    const unsigned long long mask = 0ull - br;
    Fe<M> r;
    unsigned long long w;
    w = __builtin_addcll(t[0], Pm[0] & mask, 0, &c); r.v[0] = (uint32_t)w; r.v[1] = (uint32_t)(w >> 32);
    w = __builtin_addcll(t[1], Pm[1] & mask, c, &c); r.v[2] = (uint32_t)w; r.v[3] = (uint32_t)(w >> 32);
    w = __builtin_addcll(t[2], Pm[2] & mask, c, &c); r.v[4] = (uint32_t)w; r.v[5] = (uint32_t)(w >> 32);
    w = __builtin_addcll(t[3], Pm[3] & mask, c, &c); r.v[6] = (uint32_t)w; r.v[7] = (uint32_t)(w >> 32);
    return r;
}
#endif

template <class M>
PORLA_HD Fe<M> fe_add(const Fe<M>& a, const Fe<M>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return fe_add_host64<M>(a, b);
#endif
    uint32_t t[8], s[8];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = adc32(a.v[i], b.v[i], c);
    uint32_t br = sub_p<M>(s, t);
    bool ge = (M::SPARE_BITS > 0) ? (br == 0) : (c != 0 || br == 0);
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? s[i] : t[i];
    return r;
}

template <class M>
PORLA_HD Fe<M> fe_sub(const Fe<M>& a, const Fe<M>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return fe_sub_host64<M>(a, b);
#endif
    uint32_t t[8];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = sbb32(a.v[i], b.v[i], br);
    uint32_t mask = 0u - br;  // add P back when the subtraction borrowed
    uint32_t c = 0;
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = adc32(t[i], M::P[i] & mask, c);
    return r;
}

// -a: P - a in ONE borrow chain, 0 for a = 0 -- word for word what the two-chain form (0 - a, then + P where that borrowed) gives
// for every 256-bit a, in 24 instructions instead of 38
template <class M>
PORLA_HD Fe<M> fe_neg(const Fe<M>& a) {
    Fe<M> r;
    uint32_t br = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = sbb32(M::P[i], a.v[i], br);
        nz |= a.v[i];
    }
    const uint32_t keep = nz ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] &= keep;
    return r;
}
// neg ? -a : a with the zero test folded into the select
template <class M>
PORLA_HD Fe<M> fe_neg_if(const Fe<M>& a, bool neg) {
    Fe<M> r;
    uint32_t br = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i] = sbb32(M::P[i], a.v[i], br);
        nz |= a.v[i];
    }
    const bool take = neg && nz != 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = take ? r.v[i] : a.v[i];
    return r;
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

// Product for p = 2^256 - 2^32 - FOLD on plain residues: t = a*b (16 limbs), then 2^256 = 2^32 + FOLD (mod p) twice.
template <class M>
PORLA_HD Fe<M> fe_mul_pseudo_mersenne(const Fe<M>& a, const Fe<M>& b) {
    uint32_t t[16];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PORLA_NO_ASM_MUL)
    fe_product_gfx950(a.v, b.v, t);
#else
    {
        uint64_t acc = 0;
        uint32_t acc2 = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int j = k - i;
                if (j >= 0 && j < 8) mac96(acc, acc2, a.v[i], b.v[j]);
            }
            t[k] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)acc2 << 32);
            acc2 = 0;
        }
        t[15] = (uint32_t)acc;
    }
#endif
    // s = L + H * FOLD + (H << 32),  H = t[8..15], L = t[0..7]:  10 limbs
    uint32_t s[10];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)t[8 + i] * M::FOLD + t[i] + carry;      // < 2^42 + 2^32 + 2^33
        if (i > 0) x += t[8 + i - 1];                                     // the (H << 32) term
        s[i] = (uint32_t)x;
        carry = x >> 32;
    }
    {
        uint64_t x = (uint64_t)t[15] + carry;
        s[8] = (uint32_t)x;
        s[9] = (uint32_t)(x >> 32);
    }
    // second fold: hi = s[8] + s[9] * 2^32 (< 2^34); r = s[0..7] + hi * FOLD + (hi << 32)
    const uint64_t hi = ((uint64_t)s[9] << 32) | s[8];
    const uint64_t hf = hi * M::FOLD;                                     // < 2^44
    uint32_t r[8];
    uint64_t x = (uint64_t)s[0] + (uint32_t)hf;
    r[0] = (uint32_t)x;
    x = (x >> 32) + s[1] + (uint32_t)(hf >> 32) + (uint32_t)hi;
    r[1] = (uint32_t)x;
    x = (x >> 32) + s[2] + (uint32_t)(hi >> 32);
    r[2] = (uint32_t)x;
#pragma unroll
    for (int i = 3; i < 8; i++) {
        x = (x >> 32) + s[i];
        r[i] = (uint32_t)x;
    }
    // a carry out of 2^256 is worth 2^32 + FOLD once more; the low limbs are then tiny, so this cannot carry again
    const uint32_t over = (uint32_t)(x >> 32);
    x = (uint64_t)r[0] + (over ? M::FOLD : 0u);
    r[0] = (uint32_t)x;
    x = (x >> 32) + r[1] + over;
    r[1] = (uint32_t)x;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        x = (x >> 32) + r[i];
        r[i] = (uint32_t)x;
    }
    uint32_t d[8];
    uint32_t br = sub_p<M>(d, r);
    Fe<M> out;
#pragma unroll
    for (int i = 0; i < 8; i++) out.v[i] = br ? r[i] : d[i];
    return out;
}

// the x86-64 host pass gets a hand-written product (mulx, adcx, adox); everything else (the device pass, other hosts) the portable one
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__) && (defined(__clang__) || defined(__GNUC__))
#define PORLA_FP64_ADX 1
#else
#define PORLA_FP64_ADX 0
#endif
#if PORLA_FP64_ADX
inline bool host_has_adx() {      // PORLA_NO_ADX=1 forces the portable products (the path of a host without BMI2 / ADX)
    static const bool v = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx") &&
                          !(getenv("PORLA_NO_ADX") && getenv("PORLA_NO_ADX")[0] == '1');
    return v;
}
// One round of the Montgomery product with two carry chains (adcx / adox), the modulus' top bit clear so that no sixth word is
// needed: T0..T4 += a * b[i]; m = T0 * inv; T += m * p; the round's result is (T1..T4).
#define PORLA_MM_ROUND(BI, T0, T1, T2, T3, T4)                                                  \
    "movq " BI "(%[b]), %%rdx\n\t"                                                              \
    "movq $0, " T4 "\n\t"                                                                       \
    "xorl %%eax, %%eax\n\t"                                                                     \
    "mulxq 0(%[a]), %[l], %[h]\n\t"  "adoxq %[l], " T0 "\n\t" "adcxq %[h], " T1 "\n\t"            \
    "mulxq 8(%[a]), %[l], %[h]\n\t"  "adoxq %[l], " T1 "\n\t" "adcxq %[h], " T2 "\n\t"            \
    "mulxq 16(%[a]), %[l], %[h]\n\t" "adoxq %[l], " T2 "\n\t" "adcxq %[h], " T3 "\n\t"            \
    "mulxq 24(%[a]), %[l], %[h]\n\t" "adoxq %[l], " T3 "\n\t" "adcxq %[h], " T4 "\n\t"            \
    "adoxq %%rax, " T4 "\n\t"                                                                   \
    "movq " T0 ", %%rdx\n\t"                                                                    \
    "imulq %[inv], %%rdx\n\t"                                                                   \
    "xorl %%eax, %%eax\n\t"                                                                     \
    "mulxq 0(%[p]), %[l], %[h]\n\t"  "adoxq %[l], " T0 "\n\t" "adcxq %[h], " T1 "\n\t"            \
    "mulxq 8(%[p]), %[l], %[h]\n\t"  "adoxq %[l], " T1 "\n\t" "adcxq %[h], " T2 "\n\t"            \
    "mulxq 16(%[p]), %[l], %[h]\n\t" "adoxq %[l], " T2 "\n\t" "adcxq %[h], " T3 "\n\t"            \
    "mulxq 24(%[p]), %[l], %[h]\n\t" "adoxq %[l], " T3 "\n\t" "adcxq %[h], " T4 "\n\t"            \
    "adoxq %%rax, " T4 "\n\t"
// t = a b / 2^256 mod p, below 2p (the caller subtracts p once if needed); p's top bit must be clear; needs BMI2 + ADX
__attribute__((target("bmi2,adx"))) inline void mont_mul4_adx(uint64_t t[4], const uint64_t a[4], const uint64_t b[4], const uint64_t p[4],
                                                               uint64_t inv) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, l, h;
    asm(PORLA_MM_ROUND("0", "%[t0]", "%[t1]", "%[t2]", "%[t3]", "%[t4]")
        PORLA_MM_ROUND("8", "%[t1]", "%[t2]", "%[t3]", "%[t4]", "%[t0]")
        PORLA_MM_ROUND("16", "%[t2]", "%[t3]", "%[t4]", "%[t0]", "%[t1]")
        PORLA_MM_ROUND("24", "%[t3]", "%[t4]", "%[t0]", "%[t1]", "%[t2]")
        : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [l] "=&r"(l), [h] "=&r"(h)
        : [a] "r"(a), [b] "r"(b), [p] "r"(p), [inv] "r"(inv), "m"(*(const uint64_t(*)[4])a), "m"(*(const uint64_t(*)[4])b),
          "m"(*(const uint64_t(*)[4])p)
        : "rax", "rdx", "cc");
    t[0] = t4; t[1] = t0; t[2] = t1; t[3] = t2;      // after the fourth round the result is its (T1..T4) = (t4, t0, t1, t2)
}
#undef PORLA_MM_ROUND
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
// Host pass: the same Montgomery product (radix 2^256) with 4 x 64-bit limbs and 128-bit products (CIOS) -- about 2.5x the
// speed of the 32-bit form above on x86-64; it carries every host-side tail (window fold, single-point operations, the
// pairing of verify_proof).  fe_mul_generic stays the reference the device assembly is checked against (tools/fe_check.hip).
template <class M>
inline Fe<M> fe_mul_host64(const Fe<M>& a, const Fe<M>& b) {
    typedef unsigned __int128 u128;
    static const uint64_t inv = [] {
        const uint64_t p0 = ((uint64_t)M::P[1] << 32) | M::P[0];
        uint64_t x = 1;                                   // Newton: x <- x (2 - p0 x)
        for (int i = 0; i < 6; i++) x *= 2 - p0 * x;
        return 0 - x;
    }();
    uint64_t A[4], B[4], Pm[4], t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        A[i] = ((uint64_t)a.v[2 * i + 1] << 32) | a.v[2 * i];
        B[i] = ((uint64_t)b.v[2 * i + 1] << 32) | b.v[2 * i];
        Pm[i] = ((uint64_t)M::P[2 * i + 1] << 32) | M::P[2 * i];
    }
#if PORLA_FP64_ADX
    if ((Pm[3] >> 63) == 0 && host_has_adx()) mont_mul4_adx(t, A, B, Pm, inv);     // the hand-written product (t[4] stays 0)
    else
#endif
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)A[j] * B[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * inv;
        c = (u128)m * Pm[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * Pm[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    uint64_t d[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 x = (u128)t[i] - Pm[i] - (uint64_t)br; d[i] = (uint64_t)x; br = (x >> 64) & 1; }
    const bool ge = t[4] != 0 || br == 0;
    Fe<M> r;
    for (int i = 0; i < 4; i++) { const uint64_t w = ge ? d[i] : t[i]; r.v[2 * i] = (uint32_t)w; r.v[2 * i + 1] = (uint32_t)(w >> 32); }
    return r;
}
#endif

template <class M>
PORLA_HD Fe<M> fe_mul(const Fe<M>& a, const Fe<M>& b) {
    if (M::PSEUDO_MERSENNE) return fe_mul_pseudo_mersenne<M>(a, b);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PORLA_NO_ASM_MUL)
    return fe_mul_gfx950<M>(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_generic<M>(a, b);
#else
    return fe_mul_host64<M>(a, b);
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
// t < (K + 1) P  ->  t mod P without a data-dependent loop: a quotient estimate from the top words, one product-and-subtract, one
// conditional subtraction.  qe = floor(t7 / (P7 + 1)) <= floor(t / P) <= qe + 1: the real ratios t7 / (P7 + 1) <= t / P < (t7 + 1) / P7
// differ by (t7 + P7 + 1) / (P7 (P7 + 1)) < 2^-25 for P7 >= 2^29 and t7 < 2^32, so their floors differ by at most one.
// (The ICC finish step: A mod p_icc below 4.3 q took five rounds of fe_reduce_plain's loop in every wave.)
template <class M, int K>
PORLA_HD void fe_reduce_small(uint32_t t[8]) {
    static_assert(M::P[7] >= (1u << 29), "the quotient estimate needs a modulus of more than 253 bits");
    if constexpr (K > 0 && M::P[7] != 0xffffffffu) {
        const uint32_t qe = t[7] / (M::P[7] + 1u);
        uint64_t carry = 0;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint64_t m = (uint64_t)qe * M::P[i] + carry;
            carry = m >> 32;
            t[i] = sbb32(t[i], (uint32_t)m, br);
        }
    }
    uint32_t s[8];
    const uint32_t br = sub_p<M>(s, t);
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = br ? t[i] : s[i];
}

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
