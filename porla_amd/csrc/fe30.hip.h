// Reduced-radix field arithmetic for the accumulation kernels: 9 limbs of 30 bits in 32-bit registers, Montgomery radix
// 2^270.  Same role as fe.hip.h (the reference's field layer: gnark-crypto ecc/bn254/fp behind porla/main.go:130-137) -- a
// second representation of the SAME residues, chosen for the gfx950 VALU:
//   * a 30 x 30-bit product is < 2^60, so a 64-bit column accumulator takes all 16..18 products of a column WITHOUT a
//     carry: one v_mad_u64_u32 per product and nothing else (the 8 x 32-bit form of fe.hip.h needs a v_addc per product
//     for the third accumulator word: 128 + ~120 instructions against 162 + ~60 here);
//   * the radix 2^270 exceeds p^2 / p by 2^16, so a Montgomery product of operands below 2^258 is already < p + 1:
//     no conditional subtraction, and sums / differences of a few products may stay unreduced (value bounds in the
//     comments of ec30.hip.h);
//   * the price: limbs must be < 2^30 when they enter a product (one carry ripple after every addition chain) and
//     values change representation at the kernel boundary (pack / unpack, a constant product to switch the radix).
// Overflow budget of a column accumulator (checked for every modulus by tools/check_fe30_bounds.py): operands with limbs
// 0..7 < 2^30 and limb 8 < 2^18; the widest column holds 16 products below 2^60 plus terms with a small top limb.
#pragma once
#include "fe.hip.h"

namespace porla {

constexpr uint32_t F30_MASK = (1u << 30) - 1;

template <class M>
struct P30 {
    // limb i of the modulus in radix 2^30
    static constexpr uint32_t limb(int i) {
        const int o = 30 * i, w = o / 32, s = o % 32;
        uint64_t lo = M::P[w];
        uint64_t hi = (w + 1 < 8) ? M::P[w + 1] : 0;
        return (uint32_t)(((lo | (hi << 32)) >> s) & F30_MASK);
    }
    static constexpr uint32_t neg_inv() {  // -p^-1 mod 2^30
        uint32_t p0 = limb(0), x = 1;
        for (int i = 0; i < 6; i++) x *= 2u - p0 * x;
        return (0u - x) & F30_MASK;
    }
    static constexpr uint32_t INV = neg_inv();
};

template <class M>
struct F30 {
    uint32_t v[9];
};

__device__ __forceinline__ void f30_mac(uint64_t& acc, uint32_t a, uint32_t b) { acc += (uint64_t)a * b; }
__device__ __forceinline__ void f30_mac_const(uint64_t& acc, uint32_t a, uint32_t k) { acc += (uint64_t)a * k; }

// Portable forms (what the host pass parses, and what tools/fe30_check.hip compares the generated assembly with).
// Montgomery product a * b / 2^270 mod p.  Operands: limbs 0..7 < 2^30, limb 8 < 2^18.  Result: limbs < 2^30, value
// < p + 2^246.  162 multiply-adds + 9 v_mul_lo_u32 + 17 masks + 17 shifts.
template <class M>
__device__ __forceinline__ F30<M> f30_mul_portable(const F30<M>& a, const F30<M>& b) {
    uint64_t t = 0;
    uint32_t m[9];
    F30<M> r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) f30_mac(t, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) f30_mac_const(t, m[i], P30<M>::limb(k - i));
        m[k] = ((uint32_t)t * P30<M>::INV) & F30_MASK;
        f30_mac_const(t, m[k], P30<M>::limb(0));
        t >>= 30;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) f30_mac(t, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = k - 8; i <= 8; i++) f30_mac_const(t, m[i], P30<M>::limb(k - i));
        r.v[k - 9] = (uint32_t)t & F30_MASK;
        t >>= 30;
    }
    r.v[8] = (uint32_t)t;
    return r;
}

// square: the off-diagonal products once, doubled through the operand (2 * a_i < 2^31 still fits the multiplier)
template <class M>
__device__ __forceinline__ F30<M> f30_sqr_portable(const F30<M>& a) {
    uint64_t t = 0;
    uint32_t m[9], d[9];
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) f30_mac(t, d[i], a.v[k - i]);
        if ((k & 1) == 0) f30_mac(t, a.v[k / 2], a.v[k / 2]);
#pragma unroll
        for (int i = 0; i < k; i++) f30_mac_const(t, m[i], P30<M>::limb(k - i));
        m[k] = ((uint32_t)t * P30<M>::INV) & F30_MASK;
        f30_mac_const(t, m[k], P30<M>::limb(0));
        t >>= 30;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) f30_mac(t, d[i], a.v[k - i]);
        if ((k & 1) == 0) f30_mac(t, a.v[k / 2], a.v[k / 2]);
#pragma unroll
        for (int i = k - 8; i <= 8; i++) f30_mac_const(t, m[i], P30<M>::limb(k - i));
        r.v[k - 9] = (uint32_t)t & F30_MASK;
        t >>= 30;
    }
    r.v[8] = (uint32_t)t;
    return r;
}

// (a b + c d) / 2^270 mod p with ONE reduction.  The products of c d join the accumulator of a b's product-and-reduce chain
// directly where the column still fits 64 bits; in the three widest columns (6..8: 18 + 8 full-size products) they run in a side
// chain `u` with its own carries, which hands its low 30 bits to `t` per column and its carry to column 9
// (tools/check_fe30_bounds.py: check_mul2).  Operands as for f30_mul_portable; result: limbs < 2^30, value < p + 2^247.
constexpr int F30_MUL2_CHAIN_LO = 6, F30_MUL2_CHAIN_HI = 8;
template <class M>
__device__ __forceinline__ F30<M> f30_mul2_portable(const F30<M>& a, const F30<M>& b, const F30<M>& c, const F30<M>& d) {
    uint64_t t = 0, u = 0;
    uint32_t m[9];
    F30<M> r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const bool chained = k >= F30_MUL2_CHAIN_LO && k <= F30_MUL2_CHAIN_HI;
        if (k == F30_MUL2_CHAIN_HI + 1) t += u >> 30;
        if (chained) {
            if (k > F30_MUL2_CHAIN_LO) u >>= 30;
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) f30_mac(u, c.v[i], d.v[k - i]);
            t += (uint32_t)u & F30_MASK;
        }
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) {
            f30_mac(t, a.v[i], b.v[k - i]);
            if (!chained) f30_mac(t, c.v[i], d.v[k - i]);
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) f30_mac_const(t, m[i], P30<M>::limb(k - i));
            m[k] = ((uint32_t)t * P30<M>::INV) & F30_MASK;
            f30_mac_const(t, m[k], P30<M>::limb(0));
        } else {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) f30_mac_const(t, m[i], P30<M>::limb(k - i));
            r.v[k - 9] = (uint32_t)t & F30_MASK;
        }
        t >>= 30;
    }
    r.v[8] = (uint32_t)t;
    return r;
}

// ---------------------------------------------------------------- special-form modulus p = 2^256 - 2^32 - FOLD (secp256k1)
// Plain residues (no Montgomery factor), as in fe.hip.h and in the reference's field_5x52.  The 18-limb schoolbook product is
// folded with 2^270 = 2^14 (2^32 + FOLD) = C1 2^30 + C0 (mod p), then everything above 2^256 once more with
// 2^256 = 2^32 + FOLD.  Operands: limbs 0..7 < 2^30, limb 8 < 2^19 (value < 2^259).  Result: limbs < 2^30, value < 2^256 + 2^49.
template <class M>
struct PM30 {
    static constexpr uint32_t C0 = (M::FOLD << 14) & F30_MASK;              // low limb of 2^14 (2^32 + FOLD); FOLD < 2^16
    static constexpr uint32_t C1 = (1u << 16) + ((M::FOLD << 14) >> 30);    // 2^46 / 2^30 (+ the carry of FOLD 2^14)
};
// t += a * b as ONE v_mad_u64_u32 (b: a constant in a scalar register, or the inline constant 1 for a plain 32-bit term).  Left to
// the compiler, `t += x` zero-extends x into a register pair first (a move and a 64-bit add) and a product with C1 = 2^16 + k becomes a
// 64-bit shift, a product and two adds: the fold below was 9 instructions per column, 5 this way (secp256k1: a sixth of the
// accumulation loop's instruction stream, profiles/r05_ah_*)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void f30_acc_mad(uint64_t& t, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(a), "s"(b) : "vcc");
}
__device__ __forceinline__ void f30_acc_add(uint64_t& t, uint32_t a) {
    asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(t) : "v"(a) : "vcc");
}
#else
__device__ __forceinline__ void f30_acc_mad(uint64_t& t, uint32_t a, uint32_t b) { t += (uint64_t)a * b; }
__device__ __forceinline__ void f30_acc_add(uint64_t& t, uint32_t a) { t += a; }
#endif
template <class M>
__device__ __forceinline__ void f30_pm_fold(F30<M>& r, const uint32_t (&L)[18]) {
    // phase 2: columns 0..9 of  lo + hi * (C1 2^30 + C0),  hi = L[9..17]
    uint64_t t = 0;
    uint32_t R[10];
#pragma unroll
    for (int j = 0; j < 10; j++) {
        if (j < 9) { f30_acc_add(t, L[j]); f30_acc_mad(t, L[9 + j], PM30<M>::C0); }
        if (j >= 1) f30_acc_mad(t, L[9 + j - 1], PM30<M>::C1);
        R[j] = (uint32_t)t & F30_MASK;
        t >>= 30;
    }
    // hi < 2^248, so hi * (C1 2^30 + C0) < 2^295: R[9] < 2^25 and nothing is left in t.
    // phase 3: g1 = bits 256.. of limb 8, g2 = R[9] (weight 2^270):  + g1 (2^32 + FOLD) + g2 (C1 2^30 + C0)
    const uint32_t g1 = R[8] >> 16, g2 = R[9];
    uint64_t u = (uint64_t)R[0] + (uint64_t)g1 * M::FOLD + (uint64_t)g2 * PM30<M>::C0;
    r.v[0] = (uint32_t)u & F30_MASK;
    u >>= 30;
    u += (uint64_t)R[1] + ((uint64_t)g1 << 2) + (uint64_t)g2 * PM30<M>::C1;
    r.v[1] = (uint32_t)u & F30_MASK;
    uint32_t carry = (uint32_t)(u >> 30);
#pragma unroll
    for (int j = 2; j < 8; j++) {
        uint32_t x = R[j] + carry;
        r.v[j] = x & F30_MASK;
        carry = x >> 30;
    }
    r.v[8] = (R[8] & 0xffffu) + carry;
}
// value < 2^259 (normal limbs) -> the same residue below 2^256 + 2^49: only the bits above 2^256 are folded
template <class M>
__device__ __forceinline__ F30<M> f30_pm_reduce(const F30<M>& a) {
    F30<M> r;
    const uint32_t g1 = a.v[8] >> 16;
    uint64_t u = (uint64_t)a.v[0] + (uint64_t)g1 * M::FOLD;
    r.v[0] = (uint32_t)u & F30_MASK;
    u >>= 30;
    u += (uint64_t)a.v[1] + ((uint64_t)g1 << 2);
    r.v[1] = (uint32_t)u & F30_MASK;
    uint32_t carry = (uint32_t)(u >> 30);
#pragma unroll
    for (int j = 2; j < 8; j++) {
        uint32_t x = a.v[j] + carry;
        r.v[j] = x & F30_MASK;
        carry = x >> 30;
    }
    r.v[8] = (a.v[8] & 0xffffu) + carry;
    return r;
}
template <class M>
__device__ __forceinline__ F30<M> f30_mul_pm_portable(const F30<M>& a, const F30<M>& b) {
    uint64_t t = 0;
    uint32_t L[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) t += (uint64_t)a.v[i] * b.v[k - i];
        L[k] = (uint32_t)t & F30_MASK;
        t >>= 30;
    }
    L[17] = (uint32_t)t;
    F30<M> r;
    f30_pm_fold<M>(r, L);
    return r;
}
template <class M>
__device__ __forceinline__ F30<M> f30_sqr_pm_portable(const F30<M>& a) { return f30_mul_pm_portable<M>(a, a); }
// a b + c d mod p with ONE fold (the special-form twin of f30_mul2_portable).  The double-width sum is < 2^519, its high part
// hi < 2^249: hi (C1 2^30 + C0) < 2^296, so the fold's R[9] stays below 2^26 and its result below 2^256 + 2^50 (limbs normal).
// Column 7 holds 2 x 8 full-size products -- within 2^34 of 2^64 in one accumulator -- so its c d terms go through a side chain.
constexpr int F30_PM_MUL2_CHAIN = 7;
template <class M>
__device__ __forceinline__ F30<M> f30_mul2_pm_portable(const F30<M>& a, const F30<M>& b, const F30<M>& c, const F30<M>& d) {
    uint64_t t = 0, u = 0;
    uint32_t L[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        if (k == F30_PM_MUL2_CHAIN + 1) t += u >> 30;
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) {
            t += (uint64_t)a.v[i] * b.v[k - i];
            if (k == F30_PM_MUL2_CHAIN) u += (uint64_t)c.v[i] * d.v[k - i];
            else t += (uint64_t)c.v[i] * d.v[k - i];
        }
        if (k == F30_PM_MUL2_CHAIN) t += (uint32_t)u & F30_MASK;
        L[k] = (uint32_t)t & F30_MASK;
        t >>= 30;
    }
    L[17] = (uint32_t)t;
    F30<M> r;
    f30_pm_fold<M>(r, L);
    return r;
}

// The device forms: generated assembly blocks (tools/gen_fe30_asm.py), 205 / 169 instructions per product / square.
#if defined(__HIP_DEVICE_COMPILE__)
#include "fe30_mul_gfx950.inc"
#else
template <class M>
__device__ __forceinline__ F30<M> f30_mul_mont(const F30<M>& a, const F30<M>& b) { return f30_mul_portable<M>(a, b); }
template <class M>
__device__ __forceinline__ F30<M> f30_sqr_mont(const F30<M>& a) { return f30_sqr_portable<M>(a); }
template <class M>
__device__ __forceinline__ F30<M> f30_mul_pm(const F30<M>& a, const F30<M>& b) { return f30_mul_pm_portable<M>(a, b); }
template <class M>
__device__ __forceinline__ F30<M> f30_sqr_pm(const F30<M>& a) { return f30_sqr_pm_portable<M>(a); }
template <class M>
__device__ __forceinline__ F30<M> f30_mul_icc(const F30<M>& a, const F30<M>& b) { return f30_mul_portable<M>(a, b); }
template <class M>
__device__ __forceinline__ F30<M> f30_mul2_mont(const F30<M>& a, const F30<M>& b, const F30<M>& c, const F30<M>& d) {
    return f30_mul2_portable<M>(a, b, c, d);
}
template <class M>
__device__ __forceinline__ F30<M> f30_mul_mont_tied(const F30<M>& a, const F30<M>& b) { return f30_mul_portable<M>(a, b); }
template <class M>
__device__ __forceinline__ F30<M> f30_mul2_pm(const F30<M>& a, const F30<M>& b, const F30<M>& c, const F30<M>& d) {
    return f30_mul2_pm_portable<M>(a, b, c, d);
}
#endif
// a b + c d with one reduction / one fold
template <class M>
__device__ __forceinline__ F30<M> f30_mul2(const F30<M>& a, const F30<M>& b, const F30<M>& c, const F30<M>& d) {
    if constexpr (M::PSEUDO_MERSENNE) return f30_mul2_pm<M>(a, b, c, d);
    else return f30_mul2_mont<M>(a, b, c, d);
}
template <class M>
__device__ __forceinline__ F30<M> f30_mul(const F30<M>& a, const F30<M>& b) {
    if constexpr (M::PSEUDO_MERSENNE) return f30_mul_pm<M>(a, b);
    else return f30_mul_mont<M>(a, b);
}
template <class M>
__device__ __forceinline__ F30<M> f30_sqr(const F30<M>& a) {
    if constexpr (M::PSEUDO_MERSENNE) return f30_sqr_pm<M>(a);
    else return f30_sqr_mont<M>(a);
}

// 8 x 32-bit words (a value < 2^256) -> 9 x 30-bit limbs
template <class M>
__device__ __forceinline__ F30<M> f30_unpack(const uint32_t w[8]) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int o = 30 * i, q = o / 32, s = o % 32;
        uint32_t lo = w[q];
        uint32_t hi = (q + 1 < 8) ? w[q + 1] : 0;
        uint32_t x = s == 0 ? lo : (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
        r.v[i] = (i < 8) ? (x & F30_MASK) : x;
    }
    return r;
}
// 9 x 30-bit limbs (limbs < 2^30, value < 2^256) -> 8 x 32-bit words
template <class M>
__device__ __forceinline__ void f30_pack(uint32_t w[8], const F30<M>& a) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        // word q = bits [32q, 32q+32): from limbs floor(32q/30) and the next
        const int i = (32 * q) / 30, s = 32 * q - 30 * i;      // bit s of limb i is bit 0 of the word
        uint64_t x = (uint64_t)a.v[i] >> s;
        x |= (uint64_t)a.v[i + 1] << (30 - s);
        if (i + 2 < 9 && 60 - s < 32) x |= (uint64_t)a.v[i + 2] << (60 - s);
        w[q] = (uint32_t)x;
    }
}

// ---------------------------------------------------------------- additions and subtractions
// Products need NORMAL operands: limbs 0..7 < 2^30, limb 8 < 2^18.  Limb-wise sums are rippled back to that form before
// they enter a product; values are tracked in multiples of p in the comments of ec30.hip.h (everything stays below 8p).
template <class M>
__device__ __forceinline__ void f30_ripple(F30<M>& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.v[i + 1] += a.v[i] >> 30;
        a.v[i] &= F30_MASK;
    }
}

// K * p written with limbs large enough for a borrow-free limb-wise "+ K p - b" of a NORMAL b whose value is at most
// (K - 1) p + 2^240: limb 0 gets + 2^30, limbs 1..7 get + 2^30 - 1, limb 8 gets - 1 (the sum is still K p).
template <class M, int K>
struct KP30 {
    struct Tab { uint32_t v[9]; };
    static constexpr Tab make() {
        Tab t{};
        uint64_t carry = 0;
        for (int i = 0; i < 9; i++) {
            uint64_t x = (uint64_t)K * P30<M>::limb(i) + carry;
            t.v[i] = (uint32_t)(x & F30_MASK);
            carry = x >> 30;
        }
        t.v[8] += (uint32_t)(carry << 30);   // K p < 2^258: nothing above limb 8
        t.v[0] += 1u << 30;
        for (int i = 1; i < 8; i++) t.v[i] += (1u << 30) - 1;
        t.v[8] -= 1;
        return t;
    }
    static constexpr Tab T = make();
};
template <class M, int K>
constexpr typename KP30<M, K>::Tab KP30<M, K>::T;

// a - b + K p, normal result.  a normal (or any limbs < 2^30 + 2^29), b normal with value <= (K-1) p + 2^240.
template <class M, int K>
__device__ __forceinline__ F30<M> f30_sub(const F30<M>& a, const F30<M>& b) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (KP30<M, K>::T.v[i] - b.v[i]);
    f30_ripple<M>(r);
    return r;
}
// a - 2 b + K p in one pass, normal result.  a, b normal, 2 b <= (K-1) p + 2^240.  The table of K p is the one of f30_sub with
// the borrow allowance doubled (limb 0 + 2^31, limbs 1..7 + 2^31 - 2, limb 8 - 2): every limb-wise sum stays below 2^32
// (2^30 + 2^30 + 2^31 - 2) and non-negative.
template <class M, int K>
__device__ __forceinline__ F30<M> f30_sub_twice(const F30<M>& a, const F30<M>& b) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t extra = i == 0 ? (1u << 30) : (i < 8 ? (1u << 30) - 1u : 0xffffffffu);   // on top of f30_sub's own allowance
        r.v[i] = a.v[i] + ((KP30<M, K>::T.v[i] + extra) - (b.v[i] << 1));
    }
    f30_ripple<M>(r);
    return r;
}
// a + 2 b, normal result
template <class M>
__device__ __forceinline__ F30<M> f30_add2(const F30<M>& a, const F30<M>& b) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (b.v[i] << 1);
    f30_ripple<M>(r);
    return r;
}
// small multiple k * a (k <= 3), normal result
template <class M, int K>
__device__ __forceinline__ F30<M> f30_small_mul(const F30<M>& a) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] * K;
    f30_ripple<M>(r);
    return r;
}

// a product's result (normal limbs, value < p + 2^246) is 0 mod p iff it equals 0 or p
template <class M>
__device__ __forceinline__ bool f30_product_is_zero(const F30<M>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { z |= a.v[i]; e |= a.v[i] ^ P30<M>::limb(i); }
    return z == 0 || e == 0;
}

template <class M>
__device__ __forceinline__ F30<M> f30_from_fe(const Fe<M>& a) { return f30_unpack<M>(a.v); }

// canonical residue of a product's result (normal limbs; < p + 2^246 in the Montgomery form, < 2^256 + 2^49 in the special
// form, where bit 256 may be set): at most two subtractions of p
template <class M>
__device__ __forceinline__ Fe<M> f30_to_fe_canonical(const F30<M>& a) {
    uint32_t t[8], s[8];
    f30_pack<M>(t, a);                                   // the low 256 bits
    if constexpr (M::PSEUDO_MERSENNE) {
        if (a.v[8] >> 16) {                              // value = 2^256 + t: subtract p = 2^256 - (2^32 + FOLD)
            uint64_t c = (uint64_t)t[0] + M::FOLD;
            t[0] = (uint32_t)c; c >>= 32;
            c += (uint64_t)t[1] + 1u;
            t[1] = (uint32_t)c; c >>= 32;
#pragma unroll
            for (int i = 2; i < 8; i++) { c += t[i]; t[i] = (uint32_t)c; c >>= 32; }
        }
    }
    uint32_t br = sub_p<M>(s, t);
    Fe<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? t[i] : s[i];
    return r;
}
template <class M>
__device__ __forceinline__ F30<M> f30_const(const uint32_t (&w)[8]) {
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int o = 30 * i, q = o / 32, sft = o % 32;
        uint64_t lo = w[q];
        uint64_t hi = (q + 1 < 8) ? w[q + 1] : 0;
        r.v[i] = (uint32_t)(((lo | (hi << 32)) >> sft) & (i < 8 ? F30_MASK : 0xffffffffu));
    }
    return r;
}

}  // namespace porla
