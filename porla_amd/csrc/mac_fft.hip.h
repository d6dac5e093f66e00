// MAC-side ICC encode ("FFT in the exponent") on gfx950: the butterfly network of icc.hip.h applied to group elements.
//
// Reference: Server::CRebuild_Cached interleaves, with every data butterfly, the same butterfly on the block MACs
//   tm = v^j * MAC[k+m2];  MAC[k] = um + tm;  MAC[k+m2] = um - tm
// (porla/Server/Server.hpp:1590-1609 and :1658-1676, KZG branch: bn254_mult / bn254_add / bn254_neg = one cgo call
// each; IPA branch: secp256k1_ecmult + gej_add_var), after the init scaling X = MAC_U, Y = wt * MAC_U
// (Server.hpp:1523-1536).  The client runs the same network on the MAC complements (porla/Client/Client.hpp:1040-1453).
// The multiplier is the INTEGER v^j mod p_icc handed to the group as a scalar (convert_ZZ_to_scalar, utils.h:307-318),
// i.e. reduced mod the group order by fr.SetBytes (main.go:209) / the secp256k1 scalar.
//
// MI355X: N/2 independent scalar multiplications per stage, log2 N dependent stages; extended-Jacobian points (128 B per MAC, the
// lazy memory form of ec30.hip.h) stay in HBM between stages: 2 * 128 B per MAC per stage, against ~2 000 field products per
// butterfly -- VALU (integer multiply) bound.  Up to 2^16 rows a butterfly runs on the four lanes of a quad with its accumulator
// in registers (k_mac_stage30_quad), above that one lane per butterfly (k_mac_stage30).
#pragma once
#include "fixed_base.hip.h"
#include "icc.hip.h"

namespace porla {

// a 256-bit scalar handed to a kernel BY VALUE (the init scaling wt of the Y part): no device buffer to fill, so no copy the
// host would have to wait for before it may reuse its temporary
struct MacScalar { uint32_t v[8]; };

// ---- width-5 non-adjacent form of a 128-bit magnitude, one position per call (least significant first); used where a whole wave
// multiplies by ONE scalar, so that control flow is the wave's and a zero digit costs nothing (macq_ladder_uniform,
// mac30_scalar_mul_uniform).  Code of a position: 0 = zero digit, else 16 | sign << 3 | (|d| - 1) / 2 (index into the table of odd
// multiples P, 3P, .. 15P); `flip` folds the half-scalar's own sign in.
constexpr int MACQ_WNAF_LEN = 129;                // digits of a width-5 NAF of a 128-bit magnitude
__device__ __forceinline__ uint32_t mac_wnaf5_step(uint32_t (&k)[5], bool flip) {
    uint32_t code = 0;
    if (k[0] & 1u) {
        const uint32_t low = k[0] & 31u;
        const bool negd = low >= 16u;                                       // digit = low - 32
        const uint32_t mag = negd ? 32u - low : low;
        code = 16u | ((negd != flip) ? 8u : 0u) | ((mag - 1u) >> 1);
        // k -= digit: clears the low five bits; a negative digit carries 32 upwards
        uint64_t t = (uint64_t)(k[0] & ~31u) + (negd ? 32u : 0u);
        k[0] = (uint32_t)t;
#pragma unroll
        for (int i = 1; i < 5; i++) { t = (uint64_t)k[i] + (t >> 32); k[i] = (uint32_t)t; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) k[i] = (k[i] >> 1) | (k[i + 1] << 31);
    k[4] >>= 1;
    return code;
}


// plain little-endian limbs of (w^e mod p_icc) mod q for e in [0, n)   (cf. k_icc_twiddles in icc.hip.h)
// The recoding of a TWIDDLE does not depend on the butterfly: the stages whose waves share a twiddle (>= 16 butterflies per twiddle
// on four / eight lanes, >= 64 on one) read their 129 digit codes from a table made once per (N, curve) beside the twiddles
// themselves -- entry t = the codes of exponent 32 t (those stages only use exponents that are multiples of 32), low byte k1, high
// byte k2, MACQ_CODES_STRIDE 16-bit words per entry.  In the kernel the endomorphism split and the 2 x 129 recoding steps were ~10 k
// scalar instructions per wave and stage: 6-8 % of a uniform stage's time (SQ counters, profiles/r05_w_*).
constexpr int MACQ_CODES_STRIDE = 132;            // 16-bit words per entry (264 bytes: 8-byte aligned)
constexpr int MACQ_CODES_EXP_SHIFT = 5;           // entry = exponent >> 5
template <class C>
__global__ void k_mac_wnaf_codes(const uint32_t* __restrict__ tws, uint32_t entries, uint16_t* __restrict__ codes) {
    using G = typename C::Glv;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= entries) return;
    uint32_t sc[8];
    const uint4* w4 = reinterpret_cast<const uint4*>(tws + ((size_t)t << MACQ_CODES_EXP_SHIFT) * 8);
    const uint4 a = w4[0], b = w4[1];
    sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    uint32_t m0[4], m1[4];
    bool ng0, ng1;
    glv_split<G>(sc, m0, ng0, m1, ng1);
    uint32_t k0[5] = {m0[0], m0[1], m0[2], m0[3], 0u}, k1[5] = {m1[0], m1[1], m1[2], m1[3], 0u};
    uint16_t* dst = codes + (size_t)t * MACQ_CODES_STRIDE;
#pragma unroll 1
    for (int i = 0; i < MACQ_WNAF_LEN; i++) dst[i] = (uint16_t)(mac_wnaf5_step(k0, ng0) | (mac_wnaf5_step(k1, ng1) << 8));
    for (int i = MACQ_WNAF_LEN; i < MACQ_CODES_STRIDE; i++) dst[i] = 0;
}
// a wave copies its twiddle's entry into its LDS code array (66 32-bit words: two rounds of the wave); ends with the wave's fence
__device__ __forceinline__ void mac_codes_to_lds(uint16_t* lds_codes, const uint16_t* __restrict__ entry, uint32_t lane) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(entry);
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds_codes);
    for (uint32_t i = lane; i < MACQ_CODES_STRIDE / 2; i += 64) dst[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <class Q>
__global__ void k_mac_twiddles(uint32_t* __restrict__ tws, uint32_t n, const Fe<IccFp>* __restrict__ wpow, int logn) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    Fe<IccFp> acc = fe_one<IccFp>();
    for (int i = 0; i < logn; i++) {
        if ((e >> i) & 1) acc = fe_mul_call<IccFp>(acc, wpow[i]);
    }
    Fe<IccFp> plain = fe_from_mont<IccFp>(acc);
    uint32_t t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = plain.v[k];
    fe_reduce_plain<Q>(t, Q::MAX_Q_P + 1);
    uint4* d = reinterpret_cast<uint4*>(tws + (size_t)e * 8);
    d[0] = make_uint4(t[0], t[1], t[2], t[3]);
    d[1] = make_uint4(t[4], t[5], t[6], t[7]);
}

// ---------------------------------------------------------------- the scalar-multiplication ladder, one lane per point (large N)
// k * P with the curve's endomorphism: k = k1 + lambda k2 (glv.hip.h; |k1|, |k2| < 2^128), then ONE joint ladder over both
// halves -- 33 signed 4-bit windows, 4 doublings each, and per window at most one addition from the table of P's multiples
// and one from the same table with X scaled by beta (= the table of phi(P)): 132 doublings + <= 66 additions instead of
// 260 + 65, every field product in the 9 x 30-bit form of fe30.hip.h.  Points are in the lazy memory form of ec30.hip.h.
template <class C>
__device__ __noinline__ void mac30_scalar_mul(XYZZ<typename C::Fp>* out, const XYZZ<typename C::Fp>* P, const uint32_t k[8]) {
    using M = typename C::Fp;
    using G = typename C::Glv;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) nz |= k[i];
    uint4* o4 = reinterpret_cast<uint4*>(out);
    const uint4* p4 = reinterpret_cast<const uint4*>(P);
    if (nz == 0 && k[0] <= 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) o4[i] = k[0] ? p4[i] : make_uint4(0, 0, 0, 0);
        return;
    }
    uint32_t m[2][4];
    bool ng[2];
    glv_split<G>(k, m[0], ng[0], m[1], ng[1]);
    // beta in the 2^270 form
    Fe<M> bplain, r2;
#pragma unroll
    for (int i = 0; i < 8; i++) { bplain.v[i] = G::BETA[i]; r2.v[i] = M::R2_30[i]; }
    const F30<M> beta30 = f30_from_fe<M>(fe_mul_call<M>(bplain, r2));
    XYZZ<M> tbl[8];
    {
        uint4* t4 = reinterpret_cast<uint4*>(&tbl[0]);
#pragma unroll
        for (int i = 0; i < 8; i++) t4[i] = p4[i];
    }
#pragma unroll 1
    for (int i = 1; i < 8; i++) {
        const uint4* s4 = reinterpret_cast<const uint4*>(&tbl[i - 1]);
        uint4* d4 = reinterpret_cast<uint4*>(&tbl[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) d4[j] = s4[j];
        xyzz30_add_mem<M>(&tbl[i], P, 0, 0, &beta30);
    }
    // signed digits, least significant first: d in [-8, 8]; 32 windows cover 128 bits, the 33rd takes the carry
    int8_t dig[2][33];
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        uint32_t carry = 0;
#pragma unroll 1
        for (int i = 0; i < 32; i++) {
            uint32_t limb = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) limb = (j == (i >> 3)) ? m[h][j] : limb;
            uint32_t d = ((limb >> ((i & 7) * 4)) & 15u) + carry;
            if (d > 8) { dig[h][i] = (int8_t)((int)d - 16); carry = 1; }
            else { dig[h][i] = (int8_t)d; carry = 0; }
        }
        dig[h][32] = (int8_t)carry;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) o4[i] = make_uint4(0, 0, 0, 0);      // infinity
#pragma unroll 1
    for (int i = 32; i >= 0; i--) {
        xyzz30_double_mem<M>(out, 4);
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const int d = dig[h][i];
            if (d != 0) xyzz30_add_mem<M>(out, &tbl[(d < 0 ? -d : d) - 1], (uint32_t)((d < 0) != ng[h]), (uint32_t)h, &beta30);
        }
    }
}

// The same product for a scalar that is THE SAME ON ALL 64 LANES of the wave (the stages of a large network where >= 64 butterflies
// share a twiddle; the init scaling): width-5 NAF of the two half-scalars over the table of odd multiples -- 128 doublings + ~43
// additions instead of 132 + ~62 (1 910 field products instead of 2 210) -- with the recoding on the scalar unit, its 129 codes in
// `codes` (LDS, this wave's own), and runs of zero digits doubled in one call.
template <class C>
__device__ __noinline__ void mac30_scalar_mul_uniform(XYZZ<typename C::Fp>* out, const XYZZ<typename C::Fp>* P, const uint32_t k[8],
                                                      uint16_t* codes, const uint16_t* __restrict__ pre = nullptr) {
    using M = typename C::Fp;
    using G = typename C::Glv;
    if (pre) {                                                            // the twiddle's entry of the code table (k is not read)
        mac_codes_to_lds(codes, pre, threadIdx.x & 63u);
    } else {
        uint32_t m0[4], m1[4];
        bool ng0, ng1;
        glv_split<G>(k, m0, ng0, m1, ng1);
        uint32_t k0[5] = {m0[0], m0[1], m0[2], m0[3], 0u}, k1[5] = {m1[0], m1[1], m1[2], m1[3], 0u};
#pragma unroll 1
        for (int i = 0; i < MACQ_WNAF_LEN; i++) {
            const uint32_t code = mac_wnaf5_step(k0, ng0) | (mac_wnaf5_step(k1, ng1) << 8);
            if ((threadIdx.x & 63u) == 0u) codes[i] = (uint16_t)code;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    Fe<M> bplain, r2;
#pragma unroll
    for (int i = 0; i < 8; i++) { bplain.v[i] = G::BETA[i]; r2.v[i] = M::R2_30[i]; }
    const F30<M> beta30 = f30_from_fe<M>(fe_mul_call<M>(bplain, r2));
    uint4* o4 = reinterpret_cast<uint4*>(out);
    const uint4* p4 = reinterpret_cast<const uint4*>(P);
    XYZZ<M> tbl[8], two_p;                                                // (2 i + 1) P, and 2 P
    {
        uint4* t4 = reinterpret_cast<uint4*>(&tbl[0]);
        uint4* d4 = reinterpret_cast<uint4*>(&two_p);
#pragma unroll
        for (int i = 0; i < 8; i++) { t4[i] = p4[i]; d4[i] = p4[i]; }
    }
    xyzz30_double_mem<M>(&two_p, 1);
#pragma unroll 1
    for (int i = 1; i < 8; i++) {
        const uint4* s4 = reinterpret_cast<const uint4*>(&tbl[i - 1]);
        uint4* d4 = reinterpret_cast<uint4*>(&tbl[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) d4[j] = s4[j];
        xyzz30_add_mem<M>(&tbl[i], &two_p, 0, 0, &beta30);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) o4[i] = make_uint4(0, 0, 0, 0);          // infinity
    int run = 0;
#pragma unroll 1
    for (int i = MACQ_WNAF_LEN - 1; i >= 0; i--) {
        const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane((int)codes[i]);
        run++;
        if (code == 0u) continue;
        xyzz30_double_mem<M>(out, run);                                   // (an accumulator at infinity stays where it is)
        run = 0;
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const uint32_t cd = (code >> (8 * h)) & 0xffu;
            if (cd) xyzz30_add_mem<M>(out, &tbl[cd & 7u], (cd >> 3) & 1u, (uint32_t)h, &beta30);
        }
    }
    if (run) xyzz30_double_mem<M>(out, run);
}

// 64-byte big-endian affine point -> lazy memory form (residues in the 2^270 form; infinity = all zero)
template <class M>
__device__ __forceinline__ XYZZ<M> load_affine_be_lazy(const uint8_t* src) {
    Affine<M> a;
    load_be256(a.x.v, src);
    load_be256(a.y.v, src + 32);
    fe_reduce_plain<M>(a.x.v, 6);
    fe_reduce_plain<M>(a.y.v, 6);
    XYZZ<M> p;
    if (aff_is_inf<M>(a)) {
        p.x = fe_zero<M>(); p.y = p.x; p.zz = p.x; p.zzz = p.x;
    } else {
        Fe<M> r2;
#pragma unroll
        for (int j = 0; j < 8; j++) { r2.v[j] = M::R2_30[j]; p.zz.v[j] = M::R1_30[j]; }
        p.x = fe_mul_call<M>(a.x, r2);
        p.y = fe_mul_call<M>(a.y, r2);
        p.zzz = p.zz;
    }
    return p;
}

// 64-byte big-endian affine MACs -> work array in the lazy memory form; part 1 (Y): times wt.
// USE_WT is a template parameter: the plain load (the X part, and both parts of the xy form) must not carry the ladder's 226
// registers and 1.7 KB of scratch per lane -- with them a copy of 2^15 points took 0.28 ms (the scratch set-up), without 0.01.
template <class C, bool USE_WT>
__global__ void __launch_bounds__(64)
k_mac_load30(const uint8_t* __restrict__ in, uint32_t n, XYZZ<typename C::Fp>* __restrict__ work, MacScalar wt) {
    using M = typename C::Fp;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<M> p = load_affine_be_lazy<M>(in + (size_t)i * 64);
    if constexpr (USE_WT) {
        uint32_t k[8];
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = wt.v[j];
        XYZZ<M> r;
        __shared__ __align__(8) uint16_t wdig[MACQ_CODES_STRIDE];          // (64 lanes = one wave per block; one scalar for every MAC)
        mac30_scalar_mul_uniform<C>(&r, &p, k, wdig);
        p = r;
    }
    store_xyzz<M>(work + i, p);
}

// out[i] = wt * in[i] on the work array's own form (the Y part from the X part, see mac_fft.hip:mac_encode_core): one lane per MAC
template <class C>
__global__ void __launch_bounds__(64)
k_mac_scale30(const XYZZ<typename C::Fp>* __restrict__ in, uint32_t n, XYZZ<typename C::Fp>* __restrict__ out, MacScalar wt) {
    using M = typename C::Fp;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<M> p = load_xyzz<M>(in + i);
    uint32_t k[8];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = wt.v[j];
    XYZZ<M> r;
    __shared__ __align__(8) uint16_t wdig[MACQ_CODES_STRIDE];              // (64 lanes = one wave per block; one scalar for every MAC)
    mac30_scalar_mul_uniform<C>(&r, &p, k, wdig);
    store_xyzz<M>(out + i, r);
}

// UNIFORM (launched when n / 2^s >= 64 and n / 2 is a multiple of 256): the 64 lanes of a wave take butterflies of ONE twiddle index j
// (they differ in the block of the stage they belong to): mac30_scalar_mul_uniform
template <class C, bool UNIFORM>
__global__ void __launch_bounds__(256)      // (launched with 256 lanes: 21.1 against 22.2 ms at N = 2^17 with 64)
k_mac_stage30(XYZZ<typename C::Fp>* __restrict__ work, const uint32_t* __restrict__ tws, uint32_t n, int s,
              const uint16_t* __restrict__ codes) {
    using M = typename C::Fp;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    const uint32_t m2 = 1u << (s - 1);
    uint32_t j, k;
    if constexpr (UNIFORM) {
        const uint32_t rest = t >> 6;
        j = rest & (m2 - 1);
        k = ((((rest >> (s - 1)) << 6) | (t & 63u)) << s) + j;
        j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);              // (the same on all 64 lanes: tell the compiler)
    } else {
        j = t & (m2 - 1);
        k = ((t >> (s - 1)) << s) + j;
    }
    const uint32_t e = j * (n >> (s - 1));
    uint32_t sc[8];
    const uint4* q = reinterpret_cast<const uint4*>(tws + (size_t)e * 8);
    uint4 a = q[0], b = q[1];
    sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    XYZZ<M> hi = load_xyzz<M>(work + k + m2);
    XYZZ<M> tm;
    if (s == 1) tm = hi;                            // stage 1: every twiddle is w^0 = 1 (uniform over the launch): no ladder
    else if constexpr (UNIFORM) {
        __shared__ __align__(8) uint16_t wdig[4][MACQ_CODES_STRIDE];
        mac30_scalar_mul_uniform<C>(&tm, &hi, sc, wdig[threadIdx.x >> 6], codes + (size_t)(e >> MACQ_CODES_EXP_SHIFT) * MACQ_CODES_STRIDE);
    } else mac30_scalar_mul<C>(&tm, &hi, sc);
    XYZZ<M> sum = load_xyzz<M>(work + k);
    XYZZ<M> dif = sum;
    xyzz30_add_mem<M>(&sum, &tm, 0, 0, nullptr);
    xyzz30_add_mem<M>(&dif, &tm, 1, 0, nullptr);
    store_xyzz<M>(work + k, sum);
    store_xyzz<M>(work + k + m2, dif);
}

// ---------------------------------------------------------------- the butterfly stage with FOUR LANES per butterfly
// Below N = 2^16 rows a stage is one lone wave per SIMD walking a ladder of ~200 dependent group operations: its time is the
// latency of those operations (1.08 ms with one lane per butterfly).  Here every group operation runs on the four lanes of a
// quad (ec30.hip.h: 4 rounds of products per addition, 3 per doubling, instead of 14 / 9 in sequence).  The ladder is the same
// (endomorphism split, 33 signed 4-bit windows, one table of multiples 1 .. 8 of P).
// Round 5: the ACCUMULATOR LIVES IN REGISTERS (lane r of the quad holds coordinate r: xyzz30_dbl_quadreg / xyzz30_add_quadreg);
// LDS holds only what is read at random: the table of multiples (memory form) and, beside it, the table's X coordinates times
// beta -- the endomorphism's half reads its X there, so phi(d P) costs no product per addition (8 products once instead of one
// per window) -- and the sign of a digit is a subtraction on the one lane that loads Y.  Through LDS (round 4) every one of the
// ~200 operations packed, stored, fenced, re-loaded and unpacked the accumulator, and every addition staged its operand
// through a second slot (one more product round + one more round trip): 0.52 ms per stage of 2^14 butterflies; now see DESIGN.md.
// Quads follow their own control flow (a zero digit skips its addition, an accumulator at infinity its doublings): the lanes of
// a quad always branch together, which is all the quad permutes need.
constexpr int MACQ_BF = 64;                       // butterflies per block: 256 lanes = one wave on each SIMD of a compute unit

// signed 4-bit digit i (0 .. 32) of the 128-bit magnitude m: ((m >> 4i) & 15) + carry, minus 16 above 8.  The carry into
// window i is 1 exactly when the bits below it exceed 0x88..8 (the recoding with digits in (-8, 8] is unique: msm_small.hip.h).
__device__ __forceinline__ int mac_signed_digit(const uint32_t m[4], int i) {
    bool gt = false, eq = true;
#pragma unroll
    for (int q = 3; q >= 0; q--) {
        const int below = 4 * i - 32 * q;                                  // bits of limb q below the window
        const uint32_t mask = below <= 0 ? 0u : (below >= 32 ? 0xffffffffu : ((1u << below) - 1u));
        const uint32_t a = m[q] & mask, b = 0x88888888u & mask;
        gt = eq ? (a > b) : gt;
        eq = eq && (a == b);
    }
    uint32_t raw = gt ? 1u : 0u;
    if (i < 32) {
        uint32_t limb = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) limb = (q == (i >> 3)) ? m[q] : limb;
        raw += (limb >> ((i & 7) * 4)) & 15u;
    }
    return raw > 8u ? (int)raw - 16 : (int)raw;
}

// The quad-lane kernels run one wave per SIMD (LDS: one block per compute unit), so the compiler would happily spend 212 registers
// on them -- and then they fit beside nothing: next to the commitments of the same CRebuild (two 192-register waves per SIMD in the
// guest-room form of k_fb_commit) 128 registers are free.  Held to 128 they start at once there (the arithmetic is one field
// product per lane at a time).  secp256k1 has no such neighbour and a fold that wants more registers: C::MACQ_WAVES = 2 there.
#define MACQ_GUEST_ATTR __attribute__((amdgpu_waves_per_eu(C::MACQ_WAVES, C::MACQ_WAVES)))
// ... and their LDS state is DYNAMIC shared memory (sizeof(MacQuadLds<M>) at the launch, mac_fft.hip:macq_lds_bytes): with the
// state declared statically the compiler knows that one block fits a compute unit, concludes "occupancy 1" and gives the kernel
// descriptor 264 registers whatever the code uses -- the guest would not fit again
#define MACQ_LDS(L) extern __shared__ __align__(16) unsigned char macq_lds_raw[]; \
    MacQuadLds<M>& L = *reinterpret_cast<MacQuadLds<M>*>(macq_lds_raw)
// LDS state of a block of MACQ_BF quads.  Per quad: the table of multiples, the table's X coordinates times beta, 32 bytes of
// padding (consecutive quads start 32 bytes further round the banks: the 16 quads of a wave read different banks when their
// digits agree); per block: two slots for the rare general addition and the butterfly's upper input.
template <class M>
struct MacQuadLds {
    struct Quad {
        XYZZ<M> tbl[8];
        uint32_t bx[8][8];
        uint32_t pad[8];
    };
    Quad qd[MACQ_BF];
    XYZZ<M> acc[MACQ_BF], tmp[MACQ_BF], um[MACQ_BF];
    uint16_t wdig[4][MACQ_CODES_STRIDE];        // the wave-uniform ladder's digit codes, one set per wave of the block
};
// every lane of a quad moves "its" coordinate (32 bytes) of a point
template <class M>
__device__ __forceinline__ void macq_copy_coord(XYZZ<M>* dst, const XYZZ<M>* src, uint32_t r) {
    const uint4* a = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(src) + 8 * r);
    uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(dst) + 8 * r);
    const uint4 v0 = a[0], v1 = a[1];
    d[0] = v0; d[1] = v1;
}
// A quad's LDS state is private to its four lanes, which sit in one wave: LDS operations of a wave complete in order, so between a
// lane's store and another lane's load only the compiler has to be held back -- no s_barrier across the block's waves
__device__ __forceinline__ void macq_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// one residue in the memory form (BN254: packed as it is; secp256k1: canonical) at d
template <class M>
__device__ __forceinline__ void macq_store_residue(uint32_t* d, const F30<M>& v) {
    Fe<M> t;
    if constexpr (M::PSEUDO_MERSENNE) t = f30_to_fe_canonical<M>(f30_pm_reduce<M>(v));
    else f30_pack<M>(t.v, v);
    store_words8(d, t.v);
}
template <class M>
__device__ __forceinline__ F30<M> macq_load_residue(const uint32_t* s, bool* all_zero) {
    const uint4* q = reinterpret_cast<const uint4*>(s);
    const uint4 a = q[0], b = q[1];
    const uint32_t t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *all_zero = (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) == 0;
    return f30_unpack<M>(t);
}
// does any lane of this lane's quad say so?
__device__ __forceinline__ bool macq_quad_any(bool v, uint32_t lane) { return ((__ballot(v) >> (lane & 60u)) & 0xfull) != 0; }

// (c, inf) += (neg ? -1 : 1) * e, where e is a finite memory-form point whose X is read at `ex` (e's own, or the beta table's):
// the register-form addition, an accumulator at infinity (the result is the operand), and -- rare: equal x, i.e. the accumulator
// is +-e -- the ordinary one-lane addition through the two LDS slots.  All four lanes of a quad call it together.
template <class M>
__device__ __forceinline__ void macq_add(F30<M>& c, bool& inf, const XYZZ<M>* e, const uint32_t* ex, bool neg, XYZZ<M>* slot_a,
                                         XYZZ<M>* slot_b, uint32_t r, uint32_t lane) {
    if (!inf) {
        if (xyzz30_add_quadreg<M>(c, e, ex, neg, r, lane)) return;
        macq_store_residue<M>(reinterpret_cast<uint32_t*>(slot_a) + 8 * r, c);
    }
    // the operand as a point of its own: X from ex, the sign applied to Y
    bool z;
    F30<M> v = macq_load_residue<M>(r == 0u ? ex : reinterpret_cast<const uint32_t*>(e) + 8 * r, &z);
    if (neg && r == 1u) v = f30_sub<M, 4>(F30<M>{}, v);
    if (inf) { c = v; inf = false; return; }
    macq_store_residue<M>(reinterpret_cast<uint32_t*>(slot_b) + 8 * r, v);
    macq_sync();
    if (r == 0u) xyzz30_add_one_lane<M>(slot_a, slot_b, slot_a, false);
    macq_sync();
    c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(slot_a) + 8 * r, &z);
    inf = macq_quad_any(z && r == 2u, lane);
}

// (c, inf) = sc * P on the four lanes of quad q; P = L.qd[q].tbl[0] (memory form, written by this quad's lanes, macq_sync or a
// block barrier passed).  Lane r ends with coordinate r of the product; `inf` is the same on the four lanes.
template <class C>
__device__ __forceinline__ void macq_ladder(MacQuadLds<typename C::Fp>& L, uint32_t q, uint32_t r, uint32_t lane, const uint32_t sc[8],
                                            F30<typename C::Fp>& c, bool& inf) {
    using M = typename C::Fp;
    using G = typename C::Glv;
    typename MacQuadLds<M>::Quad& Q = L.qd[q];
    inf = true;
    bool z;
    c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[0]) + 8 * r, &z);
    if (macq_quad_any(z && r == 2u, lane)) return;                         // P is infinity: so is every multiple
    uint32_t m0[4], m1[4];
    bool ng0, ng1;
    glv_split<G>(sc, m0, ng0, m1, ng1);
    // tbl[i] = (i + 1) P: one doubling, six additions of P
    bool tinf = false;
    xyzz30_dbl_quadreg<M>(c, r);
    macq_store_residue<M>(reinterpret_cast<uint32_t*>(&Q.tbl[1]) + 8 * r, c);
#pragma unroll 1
    for (int i = 2; i < 8; i++) {
        macq_add<M>(c, tinf, &Q.tbl[0], reinterpret_cast<const uint32_t*>(&Q.tbl[0]), false, &L.acc[q], &L.tmp[q], r, lane);
        macq_store_residue<M>(reinterpret_cast<uint32_t*>(&Q.tbl[i]) + 8 * r, c);
    }
    macq_sync();
    // beta * X of the eight entries: two per lane
    {
        const F30<M> beta30 = f30_const<M>(G::BETA_30);
#pragma unroll 1
        for (int t = 0; t < 2; t++) {
            const uint32_t e = r + 4u * (uint32_t)t;
            const F30<M> x = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[e]), &z);
            macq_store_residue<M>(&Q.bx[e][0], f30_mul<M>(x, beta30));
        }
    }
    macq_sync();
#pragma unroll 1
    for (int i = 32; i >= 0; i--) {
        if (!inf) {
#pragma unroll 1
            for (int d = 0; d < 4; d++) xyzz30_dbl_quadreg<M>(c, r);
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            uint32_t mh[4];
#pragma unroll
            for (int j = 0; j < 4; j++) mh[j] = h ? m1[j] : m0[j];
            const int dg = mac_signed_digit(mh, i);
            if (dg == 0) continue;
            const uint32_t mag = (uint32_t)(dg < 0 ? -dg : dg) - 1u;
            const XYZZ<M>* e = &Q.tbl[mag];
            macq_add<M>(c, inf, e, h ? &Q.bx[mag][0] : reinterpret_cast<const uint32_t*>(e), (dg < 0) != (h ? ng1 : ng0), &L.acc[q],
                        &L.tmp[q], r, lane);
        }
    }
}
// ---- the ladder when all 16 quads of a wave multiply by the SAME scalar (stages with >= 16 butterflies per twiddle: all but the
// last four of a network).  Control flow is then the wave's, so a zero digit really costs nothing and the recoding can be sparse:
// width-5 non-adjacent form of the two half-scalars over the table of ODD multiples P, 3P, .. 15P -- 128 doublings + ~43
// additions instead of 132 + ~62 (a width-5 NAF with per-quad scalars would make the wave execute an addition at nearly every
// bit: one quad's digit is everybody's instruction stream).  The scalar work (endomorphism split, recoding) is uniform too: the
// compiler keeps it on the scalar unit; its 129 digit codes go through LDS, one set per wave.
//   code of a position = low byte for k1, high byte for k2: 0 = zero digit, else 16 | sign << 3 | (|d| - 1) / 2
template <class C>
__device__ __forceinline__ void macq_ladder_uniform(MacQuadLds<typename C::Fp>& L, uint32_t q, uint32_t r, uint32_t lane, uint32_t wave,
                                                    const uint32_t sc[8], F30<typename C::Fp>& c, bool& inf,
                                                    const uint16_t* __restrict__ pre = nullptr) {
    using M = typename C::Fp;
    using G = typename C::Glv;
    typename MacQuadLds<M>::Quad& Q = L.qd[q];
    // the digit codes first: the twiddle's entry of the code table (stages), or recoded here on the scalar unit (the init scaling)
    if (pre) {
        mac_codes_to_lds(L.wdig[wave], pre, lane);
    } else {
        uint32_t m0[4], m1[4];
        bool ng0, ng1;
        glv_split<G>(sc, m0, ng0, m1, ng1);
        uint32_t k0[5] = {m0[0], m0[1], m0[2], m0[3], 0u}, k1[5] = {m1[0], m1[1], m1[2], m1[3], 0u};
#pragma unroll 1
        for (int i = 0; i < MACQ_WNAF_LEN; i++) {
            const uint32_t code = mac_wnaf5_step(k0, ng0) | (mac_wnaf5_step(k1, ng1) << 8);
            if (lane == 0u) L.wdig[wave][i] = (uint16_t)code;
        }
    }
    inf = true;
    bool z;
    c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[0]) + 8 * r, &z);
    const bool p_inf = macq_quad_any(z && r == 2u, lane);                  // P is infinity: so is every multiple
    if (!p_inf) {
        // tbl[i] = (2 i + 1) P: 2P (parked in the beta table's space), then seven additions of it
        XYZZ<M>* two_p = reinterpret_cast<XYZZ<M>*>(&Q.bx[0][0]);
        const F30<M> p_r = c;
        xyzz30_dbl_quadreg<M>(c, r);
        macq_store_residue<M>(reinterpret_cast<uint32_t*>(two_p) + 8 * r, c);
        macq_sync();
        c = p_r;
        bool tinf = false;
#pragma unroll 1
        for (int i = 1; i < 8; i++) {
            macq_add<M>(c, tinf, two_p, reinterpret_cast<const uint32_t*>(two_p), false, &L.acc[q], &L.tmp[q], r, lane);
            macq_store_residue<M>(reinterpret_cast<uint32_t*>(&Q.tbl[i]) + 8 * r, c);
        }
        macq_sync();
        const F30<M> beta30 = f30_const<M>(G::BETA_30);
#pragma unroll 1
        for (int t = 0; t < 2; t++) {
            const uint32_t e = r + 4u * (uint32_t)t;
            const F30<M> x = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[e]), &z);
            macq_store_residue<M>(&Q.bx[e][0], f30_mul<M>(x, beta30));
        }
    }
    macq_sync();
    if (p_inf) return;
#pragma unroll 1
    for (int i = MACQ_WNAF_LEN - 1; i >= 0; i--) {
        const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.wdig[wave][i]);
        if (!inf) xyzz30_dbl_quadreg<M>(c, r);
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const uint32_t cd = (code >> (8 * h)) & 0xffu;
            if (cd == 0u) continue;
            const XYZZ<M>* e = &Q.tbl[cd & 7u];
            macq_add<M>(c, inf, e, h ? &Q.bx[cd & 7u][0] : reinterpret_cast<const uint32_t*>(e), (cd & 8u) != 0u, &L.acc[q], &L.tmp[q], r, lane);
        }
    }
}
// the ladder's result as a memory-form point at dst (LDS or global); all four lanes
template <class M>
__device__ __forceinline__ void macq_store_point(XYZZ<M>* dst, const F30<M>& c, bool inf, uint32_t r) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst) + 8 * r;
    if (inf) {
        uint4* d4 = reinterpret_cast<uint4*>(d);
        d4[0] = make_uint4(0, 0, 0, 0); d4[1] = d4[0];
    } else {
        macq_store_residue<M>(d, c);
    }
}
// the butterfly's two outputs  lo = um + tm,  hi = um - tm  from tm in registers (c, inf) and um in LDS (memory form); stored
// (memory form) at lo / hi when `live`
template <class M>
__device__ __forceinline__ void macq_butterfly_out(const XYZZ<M>* um, XYZZ<M>* slot_a, XYZZ<M>* slot_b, const F30<M>& c, bool inf,
                                                   XYZZ<M>* lo, XYZZ<M>* hi, bool live, uint32_t r, uint32_t lane) {
    bool z;
    const F30<M> u = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(um) + 8 * r, &z);
    const bool um_inf = macq_quad_any(z && r == 2u, lane);
    if (inf) {                                                             // tm = infinity: both outputs are um
        if (live) { macq_store_point<M>(lo, u, um_inf, r); macq_store_point<M>(hi, u, um_inf, r); }
        return;
    }
    const F30<M> cn = r == 1u ? f30_sub<M, 4>(F30<M>{}, c) : c;            // -tm
    if (um_inf) {
        if (live) { macq_store_point<M>(lo, c, false, r); macq_store_point<M>(hi, cn, false, r); }
        return;
    }
    F30<M> s = c;
    bool s_inf = false;
    macq_add<M>(s, s_inf, um, reinterpret_cast<const uint32_t*>(um), false, slot_a, slot_b, r, lane);
    if (live) macq_store_point<M>(lo, s, s_inf, r);
    s = cn;
    s_inf = false;
    macq_add<M>(s, s_inf, um, reinterpret_cast<const uint32_t*>(um), false, slot_a, slot_b, r, lane);
    if (live) macq_store_point<M>(hi, s, s_inf, r);
}

// UNIFORM (launched when n / 2^s >= 16): the 16 quads of a wave take butterflies of ONE twiddle index j (they differ in the block
// of the stage they belong to), so the whole wave multiplies by one scalar: macq_ladder_uniform
template <class C, bool UNIFORM>
__global__ void __launch_bounds__(4 * MACQ_BF) MACQ_GUEST_ATTR
k_mac_stage30_quad(XYZZ<typename C::Fp>* __restrict__ work, const uint32_t* __restrict__ tws, uint32_t n, int s,
                   const uint16_t* __restrict__ codes) {
    using M = typename C::Fp;
    MACQ_LDS(L);
    // a stage is one wave per SIMD walking ~200 dependent group operations: when another kernel shares the chip (the
    // commitments of the same CRebuild on a second stream) this wave must win the SIMD's issue arbitration every time it is
    // ready -- the wide kernel's waves fill the cycles in between (beside k_fb_commit a stage took 1.45 ms without this, 0.9 alone)
    __builtin_amdgcn_s_setprio(3);
    const uint32_t q = threadIdx.x >> 2, r = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint32_t t = blockIdx.x * MACQ_BF + q;
    const bool valid = t < n / 2;
    if (!valid) t = 0;                                                     // padding quads compute butterfly 0 and store nothing
    const uint32_t m2 = 1u << (s - 1);
    uint32_t j, k;
    if constexpr (UNIFORM) {
        // butterfly t = (rest, i16): j from the bits the wave shares, the stage's block number from the rest and the quad's place
        const uint32_t rest = t >> 4;
        j = rest & (m2 - 1);
        k = ((((rest >> (s - 1)) << 4) | (t & 15u)) << s) + j;
        j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);              // (the same on all 64 lanes: tell the compiler)
    } else {
        j = t & (m2 - 1);
        k = ((t >> (s - 1)) << s) + j;
    }
    const uint32_t e = j * (n >> (s - 1));
    uint32_t sc[8];
    {
        const uint4* w4 = reinterpret_cast<const uint4*>(tws + (size_t)e * 8);
        const uint4 a = w4[0], b = w4[1];
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    }
    macq_copy_coord<M>(&L.qd[q].tbl[0], work + k + m2, r);
    macq_copy_coord<M>(&L.um[q], work + k, r);
    macq_sync();
    F30<M> c;
    bool inf;
    if constexpr (UNIFORM) macq_ladder_uniform<C>(L, q, r, lane, threadIdx.x >> 6, sc, c, inf,
                                                   codes + (size_t)(e >> MACQ_CODES_EXP_SHIFT) * MACQ_CODES_STRIDE);
    else macq_ladder<C>(L, q, r, lane, sc, c, inf);
    // MAC[k] = um + tm, MAC[k + m2] = um - tm
    macq_butterfly_out<M>(&L.um[q], &L.acc[q], &L.tmp[q], c, inf, work + k, work + k + m2, valid, r, lane);
}

template <class M>
__device__ __forceinline__ void store_affine_be(uint8_t* dst, const XYZZ<M>& p);     // below

// ---------------------------------------------------------------- EIGHT lanes per butterfly (launches that leave half the chip idle)
// 2^13 butterflies and fewer (a stage of N <= 2^14 rows, Server::mix on two arrays of 2^12) give the quad-lane kernels at most one
// wave on every other SIMD: lanes are free, latency is everything.  The endomorphism split k P = k1 P + k2 phi(P) is two INDEPENDENT
// 128-bit ladders: here an OCTET owns a butterfly -- quad 0 walks k1 over the table of P's multiples, quad 1 walks k2 over the same
// table with the beta-scaled X coordinates, each with its own accumulator in registers (33 windows of 4 doublings + ONE addition
// instead of two); quad 1 hands its sum over through LDS, quad 0 adds it, hands tm back, and the two quads form one output each
// (um + tm | um - tm).  ~570 product rounds in sequence instead of ~700.  The two quads of an octet sit in one wave: the LDS traffic
// between them needs no block barrier (macq_sync).  32 octets per block: 68 KiB of LDS, two blocks per compute unit.
constexpr int MACO_BF = 32;
// (these kernels only run where the chip is under-filled: no neighbour to leave registers to -- two waves per SIMD, 256 registers)
#define MACO_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
template <class M>
struct MacOctLds {
    typename MacQuadLds<M>::Quad qd[MACO_BF];   // per octet: table of multiples + beta-scaled X
    XYZZ<M> acc[2 * MACO_BF], tmp[2 * MACO_BF];  // per quad: the slots of the rare general addition
    XYZZ<M> um[MACO_BF], xch[MACO_BF];           // per octet: the butterfly's upper input; what the quads hand each other
    uint16_t wdig[4][MACQ_CODES_STRIDE];         // k_mac_stage30_oct_uniform: the twiddle's digit codes per wave (it reads its half's byte)
};
#define MACO_LDS(L) extern __shared__ __align__(16) unsigned char macq_lds_raw[]; \
    MacOctLds<M>& L = *reinterpret_cast<MacOctLds<M>*>(macq_lds_raw)

// one half-scalar's ladder on a quad: (c, inf) = (neg ? -1 : 1) * m * T, T = the point whose multiples are in Q.tbl (phi: its image
// under the endomorphism -- X is read from Q.bx)
template <class M>
__device__ __forceinline__ void macq_ladder_half(typename MacQuadLds<M>::Quad& Q, XYZZ<M>* slot_a, XYZZ<M>* slot_b, const uint32_t m[4],
                                                 bool neg, bool phi, uint32_t r, uint32_t lane, F30<M>& c, bool& inf) {
    inf = true;
#pragma unroll 1
    for (int i = 32; i >= 0; i--) {
        if (!inf) {
#pragma unroll 1
            for (int d = 0; d < 4; d++) xyzz30_dbl_quadreg<M>(c, r);
        }
        const int dg = mac_signed_digit(m, i);
        if (dg == 0) continue;
        const uint32_t mag = (uint32_t)(dg < 0 ? -dg : dg) - 1u;
        const XYZZ<M>* e = &Q.tbl[mag];
        macq_add<M>(c, inf, e, phi ? &Q.bx[mag][0] : reinterpret_cast<const uint32_t*>(e), (dg < 0) != neg, slot_a, slot_b, r, lane);
    }
}
// dst = um + (c, inf) (memory form), um in LDS; all four lanes of a quad
template <class M>
__device__ __forceinline__ void macq_sum_out(const XYZZ<M>* um, XYZZ<M>* slot_a, XYZZ<M>* slot_b, F30<M> c, bool inf, XYZZ<M>* dst,
                                             bool live, uint32_t r, uint32_t lane) {
    bool z;
    const F30<M> u = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(um) + 8 * r, &z);
    const bool um_inf = macq_quad_any(z && r == 2u, lane);
    if (inf) { if (live) macq_store_point<M>(dst, u, um_inf, r); return; }
    if (!um_inf) macq_add<M>(c, inf, um, reinterpret_cast<const uint32_t*>(um), false, slot_a, slot_b, r, lane);
    if (live) macq_store_point<M>(dst, c, inf, r);
}
// The octet's whole butterfly: P = L.qd[o].tbl[0] and um = L.um[o] are in LDS (written by this octet's lanes, macq_sync passed);
// quad 0 (half = 0) leaves um + sc P at lo, quad 1 leaves um - sc P at hi (memory form, LDS or global) when `live`.
template <class C>
__device__ __forceinline__ void maco_butterfly(MacOctLds<typename C::Fp>& L, uint32_t o, uint32_t half, uint32_t r, uint32_t lane,
                                               const uint32_t sc[8], XYZZ<typename C::Fp>* lo, XYZZ<typename C::Fp>* hi, bool live) {
    using M = typename C::Fp;
    using G = typename C::Glv;
    typename MacQuadLds<M>::Quad& Q = L.qd[o];
    const uint32_t qi = 2u * o + half;                                     // this quad's slots
    F30<M> c;
    bool inf = true, z;
    c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[0]) + 8 * r, &z);
    const bool p_inf = macq_quad_any(z && r == 2u, lane);                  // (both quads read the same point: the same answer)
    if (!p_inf) {
        uint32_t m0[4], m1[4];
        bool ng0, ng1;
        glv_split<G>(sc, m0, ng0, m1, ng1);
        if (half == 0u) {                                                  // quad 0 builds the table: 2P, then 3P .. 8P
            bool tinf = false;
            xyzz30_dbl_quadreg<M>(c, r);
            macq_store_residue<M>(reinterpret_cast<uint32_t*>(&Q.tbl[1]) + 8 * r, c);
#pragma unroll 1
            for (int i = 2; i < 8; i++) {
                macq_add<M>(c, tinf, &Q.tbl[0], reinterpret_cast<const uint32_t*>(&Q.tbl[0]), false, &L.acc[qi], &L.tmp[qi], r, lane);
                macq_store_residue<M>(reinterpret_cast<uint32_t*>(&Q.tbl[i]) + 8 * r, c);
            }
        }
        macq_sync();
        {   // beta * X of the eight entries: one per lane of the octet
            const uint32_t e = 4u * half + r;
            const F30<M> x = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[e]), &z);
            macq_store_residue<M>(&Q.bx[e][0], f30_mul<M>(x, f30_const<M>(G::BETA_30)));
        }
        macq_sync();
        uint32_t mh[4];
#pragma unroll
        for (int j = 0; j < 4; j++) mh[j] = half ? m1[j] : m0[j];
        macq_ladder_half<M>(Q, &L.acc[qi], &L.tmp[qi], mh, half ? ng1 : ng0, half != 0u, r, lane, c, inf);
    }
    // quad 1 -> quad 0: k2 phi(P); quad 0: tm = k1 P + that, handed back; then one output each
    if (half) macq_store_point<M>(&L.xch[o], c, inf, r);
    macq_sync();
    if (half == 0u) {
        const F30<M> other = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&L.xch[o]) + 8 * r, &z);
        const bool other_inf = macq_quad_any(z && r == 2u, lane);
        if (!other_inf) {
            if (inf) { c = other; inf = false; }
            else macq_add<M>(c, inf, &L.xch[o], reinterpret_cast<const uint32_t*>(&L.xch[o]), false, &L.acc[qi], &L.tmp[qi], r, lane);
        }
    }
    macq_sync();                                                           // (quad 0 has read the slot before it writes it)
    if (half == 0u) macq_store_point<M>(&L.xch[o], c, inf, r);
    macq_sync();
    if (half) {
        c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&L.xch[o]) + 8 * r, &z);
        inf = macq_quad_any(z && r == 2u, lane);
        if (!inf && r == 1u) c = f30_sub<M, 4>(F30<M>{}, c);               // -tm
    }
    macq_sum_out<M>(&L.um[o], &L.acc[qi], &L.tmp[qi], c, inf, half ? hi : lo, live, r, lane);
}

template <class C>
__global__ void __launch_bounds__(8 * MACO_BF) MACO_ATTR
k_mac_stage30_oct(XYZZ<typename C::Fp>* __restrict__ work, const uint32_t* __restrict__ tws, uint32_t n, int s) {
    using M = typename C::Fp;
    MACO_LDS(L);
    __builtin_amdgcn_s_setprio(3);                                         // (as k_mac_stage30_quad: a latency-bound wave must win the issue arbitration)
    const uint32_t o = threadIdx.x >> 3, half = (threadIdx.x >> 2) & 1u, r = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint32_t t = blockIdx.x * MACO_BF + o;
    const bool valid = t < n / 2;
    if (!valid) t = 0;
    const uint32_t m2 = 1u << (s - 1);
    const uint32_t j = t & (m2 - 1);
    const uint32_t k = ((t >> (s - 1)) << s) + j;
    const uint32_t e = j * (n >> (s - 1));
    uint32_t sc[8];
    {
        const uint4* w4 = reinterpret_cast<const uint4*>(tws + (size_t)e * 8);
        const uint4 a = w4[0], b = w4[1];
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    }
    if (half) macq_copy_coord<M>(&L.um[o], work + k, r);
    else macq_copy_coord<M>(&L.qd[o].tbl[0], work + k + m2, r);
    macq_sync();
    maco_butterfly<C>(L, o, half, r, lane, sc, work + k, work + k + m2, valid);
}

// The same eight lanes per butterfly for the stages of a small network where >= 16 butterflies share a twiddle (N <= 2^14: the
// four-lane kernel leaves half the chip idle there too).  The two quads of an octet sit in DIFFERENT waves here -- waves 0 and 2 of the
// block walk k1, waves 1 and 3 walk k2 -- so that every wave multiplies by one half-scalar and keeps the sparse recoding's
// advantage (side by side in one wave the two digit streams would make the wave execute an addition whenever EITHER has a
// non-zero digit); the hand-overs between the quads are block barriers, which every wave reaches.
template <class C>
__global__ void __launch_bounds__(8 * MACO_BF) MACO_ATTR
k_mac_stage30_oct_uniform(XYZZ<typename C::Fp>* __restrict__ work, const uint32_t* __restrict__ tws, uint32_t n, int s,
                          const uint16_t* __restrict__ codes) {
    using M = typename C::Fp;
    using G = typename C::Glv;
    MACO_LDS(L);
    __builtin_amdgcn_s_setprio(3);
    const uint32_t wave = threadIdx.x >> 6, half = wave & 1u, lane = threadIdx.x & 63u, r = lane & 3u;
    const uint32_t o = (wave >> 1) * 16u + (lane >> 2);                   // octet of the block: 16 per wave pair
    const uint32_t qi = 2u * o + half;
    // n / 2 is a multiple of 32 and 16 octets share a twiddle (the launch condition): no padding octets
    const uint32_t t = blockIdx.x * MACO_BF + o;
    const uint32_t m2 = 1u << (s - 1);
    const uint32_t rest = t >> 4;
    const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)(rest & (m2 - 1)));
    const uint32_t k = ((((rest >> (s - 1)) << 4) | (t & 15u)) << s) + j;
    const uint32_t e = j * (n >> (s - 1));
    (void)tws;
    typename MacQuadLds<M>::Quad& Q = L.qd[o];
    if (half) macq_copy_coord<M>(&L.um[o], work + k, r);
    else macq_copy_coord<M>(&Q.tbl[0], work + k + m2, r);
    mac_codes_to_lds(L.wdig[wave], codes + (size_t)(e >> MACQ_CODES_EXP_SHIFT) * MACQ_CODES_STRIDE, lane);   // (this wave reads its half's byte)
    __syncthreads();
    F30<M> c;
    bool inf = true, z;
    c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[0]) + 8 * r, &z);
    const bool p_inf = macq_quad_any(z && r == 2u, lane);
    if (half == 0u && !p_inf) {                                           // tbl[i] = (2 i + 1) P: 2P (parked in the beta table's space), then + 2P
        XYZZ<M>* two_p = reinterpret_cast<XYZZ<M>*>(&Q.bx[0][0]);
        const F30<M> p_r = c;
        xyzz30_dbl_quadreg<M>(c, r);
        macq_store_residue<M>(reinterpret_cast<uint32_t*>(two_p) + 8 * r, c);
        macq_sync();
        c = p_r;
        bool tinf = false;
#pragma unroll 1
        for (int i = 1; i < 8; i++) {
            macq_add<M>(c, tinf, two_p, reinterpret_cast<const uint32_t*>(two_p), false, &L.acc[qi], &L.tmp[qi], r, lane);
            macq_store_residue<M>(reinterpret_cast<uint32_t*>(&Q.tbl[i]) + 8 * r, c);
        }
    }
    __syncthreads();
    if (!p_inf) {                                                         // beta * X of the eight entries: one per lane of the octet
        const uint32_t en = 4u * half + r;
        const F30<M> x = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&Q.tbl[en]), &z);
        macq_store_residue<M>(&Q.bx[en][0], f30_mul<M>(x, f30_const<M>(G::BETA_30)));
    }
    __syncthreads();
    if (!p_inf) {
#pragma unroll 1
        for (int i = MACQ_WNAF_LEN - 1; i >= 0; i--) {
            const uint32_t cd = ((uint32_t)__builtin_amdgcn_readfirstlane((int)L.wdig[wave][i]) >> (8u * half)) & 0xffu;
            if (!inf) xyzz30_dbl_quadreg<M>(c, r);
            if (cd == 0u) continue;
            const XYZZ<M>* en = &Q.tbl[cd & 7u];
            macq_add<M>(c, inf, en, half ? &Q.bx[cd & 7u][0] : reinterpret_cast<const uint32_t*>(en), (cd & 8u) != 0u, &L.acc[qi], &L.tmp[qi],
                        r, lane);
        }
    }
    if (half) macq_store_point<M>(&L.xch[o], c, inf, r);
    __syncthreads();
    if (half == 0u) {
        const F30<M> other = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&L.xch[o]) + 8 * r, &z);
        const bool other_inf = macq_quad_any(z && r == 2u, lane);
        if (!other_inf) {
            if (inf) { c = other; inf = false; }
            else macq_add<M>(c, inf, &L.xch[o], reinterpret_cast<const uint32_t*>(&L.xch[o]), false, &L.acc[qi], &L.tmp[qi], r, lane);
        }
        macq_sync();                                                       // (the four lanes have read the slot before any of them writes it)
        macq_store_point<M>(&L.xch[o], c, inf, r);
    }
    __syncthreads();
    if (half) {
        c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(&L.xch[o]) + 8 * r, &z);
        inf = macq_quad_any(z && r == 2u, lane);
        if (!inf && r == 1u) c = f30_sub<M, 4>(F30<M>{}, c);               // -tm
    }
    macq_sum_out<M>(&L.um[o], &L.acc[qi], &L.tmp[qi], c, inf, half ? work + k + m2 : work + k, true, r, lane);
}

// Server::mix's MAC part with eight lanes per i (k_mac_mix_quad's work): lanes 0 / 4 convert the two inputs on the way in, invert for
// the two outputs
template <class C>
__global__ void __launch_bounds__(8 * MACO_BF) MACO_ATTR
k_mac_mix_oct(const uint8_t* __restrict__ a0, const uint8_t* __restrict__ a1, uint32_t len, const uint32_t* __restrict__ tws,
              uint32_t tw_step, uint8_t* __restrict__ out, const uint8_t* __restrict__ b0, const uint8_t* __restrict__ b1,
              uint8_t* __restrict__ out_b) {
    using M = typename C::Fp;
    MACO_LDS(L);
    __builtin_amdgcn_s_setprio(3);                                         // (the data part of the same mix runs beside this kernel)
    if (blockIdx.y) { a0 = b0; a1 = b1; out = out_b; }                      // the second array pair of a mix (see k_mac_mix_quad)
    const uint32_t o = threadIdx.x >> 3, half = (threadIdx.x >> 2) & 1u, r = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint32_t i = blockIdx.x * MACO_BF + o;
    const bool valid = i < len;
    if (!valid) i = 0;
    uint32_t sc[8];
    {
        const uint4* w4 = reinterpret_cast<const uint4*>(tws + (size_t)i * tw_step * 8);
        const uint4 a = w4[0], b = w4[1];
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    }
    if (r == 0u) store_xyzz<M>(half ? &L.um[o] : &L.qd[o].tbl[0], load_affine_be_lazy<M>((half ? a0 : a1) + (size_t)i * 64));
    macq_sync();
    // the sums land in the table's entries 1 and 2 (the table is done with by then: each quad writes its own entry last)
    maco_butterfly<C>(L, o, half, r, lane, sc, &L.qd[o].tbl[1], &L.qd[o].tbl[2], true);
    macq_sync();
    if (valid && r == 0u)                                                  // the two inversions side by side (lanes 0 and 4)
        store_affine_be<M>(out + ((size_t)i + (half ? len : 0)) * 64, xyzz30_to_xyzz<M>(xyzz30_load_lazy<M>(&L.qd[o].tbl[1 + half])));
}

// Stage 1 of the network with four lanes per butterfly: every twiddle is w^0 = 1 (tm = MAC[k+1]), so the stage is its two
// additions and nothing else -- the ladder of a general stage would multiply by one.  256 lanes = 64 butterflies per block.
template <class C>
__global__ void __launch_bounds__(256) MACQ_GUEST_ATTR
k_mac_stage1_quad(XYZZ<typename C::Fp>* __restrict__ work, uint32_t n) {
    using M = typename C::Fp;
    __shared__ XYZZ<M> um[64], slot_a[64], slot_b[64];                     // 24 KiB: several blocks per compute unit
    const uint32_t q = threadIdx.x >> 2, r = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint32_t t = blockIdx.x * 64 + q;
    const bool valid = t < n / 2;
    if (!valid) t = 0;
    const uint32_t k = 2 * t;
    macq_copy_coord<M>(&um[q], work + k, r);
    bool z;
    const F30<M> c = macq_load_residue<M>(reinterpret_cast<const uint32_t*>(work + k + 1) + 8 * r, &z);
    const bool inf = macq_quad_any(z && r == 2u, lane);
    macq_sync();
    macq_butterfly_out<M>(&um[q], &slot_a[q], &slot_b[q], c, inf, work + k, work + k + 1, valid, r, lane);
}

// init scaling of the Y part (k_mac_load30 with use_wt) with four lanes per MAC: work[i] = wt * MAC[i]
// FROM_WORK: `in` is a work array (n points in the lazy memory form) instead of 64-byte big-endian affine MACs
template <class C, bool FROM_WORK = false>
__global__ void __launch_bounds__(4 * MACQ_BF) MACQ_GUEST_ATTR
k_mac_load30_quad(const uint8_t* __restrict__ in, uint32_t n, XYZZ<typename C::Fp>* __restrict__ work, MacScalar wt) {
    using M = typename C::Fp;
    MACQ_LDS(L);
    __builtin_amdgcn_s_setprio(3);                                         // (beside the commitments of a CRebuild, as the stage kernel)
    const uint32_t q = threadIdx.x >> 2, r = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint32_t i = blockIdx.x * MACQ_BF + q;
    const bool valid = i < n;
    if (!valid) i = 0;
    uint32_t sc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) sc[j] = wt.v[j];
    if (FROM_WORK) macq_copy_coord<M>(&L.qd[q].tbl[0], reinterpret_cast<const XYZZ<M>*>(in) + i, r);
    else if (r == 0u) store_xyzz<M>(&L.qd[q].tbl[0], load_affine_be_lazy<M>(in + (size_t)i * 64));
    macq_sync();
    F30<M> c;
    bool inf;
    macq_ladder_uniform<C>(L, q, r, lane, threadIdx.x >> 6, sc, c, inf);   // one scalar (a kernel argument) for every MAC: the sparse ladder
    if (valid) macq_store_point<M>(work + i, c, inf, r);
}


// Server::mix's MAC part (k_mac_mix below) with four lanes per i: the butterfly out[i] = A0[i] + v^i A1[i], out[i + len] = A0[i] -
// v^i A1[i] on 64-byte affine points -- lanes 0 / 1 convert the two inputs on the way in and invert for the two outputs
template <class C>
__global__ void __launch_bounds__(4 * MACQ_BF) MACQ_GUEST_ATTR
k_mac_mix_quad(const uint8_t* __restrict__ a0, const uint8_t* __restrict__ a1, uint32_t len, const uint32_t* __restrict__ tws,
               uint32_t tw_step, uint8_t* __restrict__ out, const uint8_t* __restrict__ b0, const uint8_t* __restrict__ b1,
               uint8_t* __restrict__ out_b) {
    using M = typename C::Fp;
    MACQ_LDS(L);
    // gridDim.y == 2: Server::mix runs this butterfly on the MAC commitments AND on the MAC alignments with the same v^i
    // (Server.hpp:1281-1318) -- the second array pair rides in the same launch (a stage this short is latency: two for the price of one)
    __builtin_amdgcn_s_setprio(3);                                         // (the data part of the same mix runs beside this kernel)
    if (blockIdx.y) { a0 = b0; a1 = b1; out = out_b; }
    const uint32_t q = threadIdx.x >> 2, r = threadIdx.x & 3u, lane = threadIdx.x & 63u;
    uint32_t i = blockIdx.x * MACQ_BF + q;
    const bool valid = i < len;
    if (!valid) i = 0;
    uint32_t sc[8];
    {
        const uint4* w4 = reinterpret_cast<const uint4*>(tws + (size_t)i * tw_step * 8);
        const uint4 a = w4[0], b = w4[1];
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    }
    if (r < 2u) store_xyzz<M>(r ? &L.um[q] : &L.qd[q].tbl[0], load_affine_be_lazy<M>((r ? a0 : a1) + (size_t)i * 64));   // lanes 0 and 1 side by side
    macq_sync();
    F30<M> c;
    bool inf;
    macq_ladder<C>(L, q, r, lane, sc, c, inf);
    macq_butterfly_out<M>(&L.um[q], &L.acc[q], &L.tmp[q], c, inf, &L.qd[q].tbl[1], &L.qd[q].tbl[2], true, r, lane);      // sum, difference (the table is done with)
    macq_sync();
    if (valid && r < 2u)                                                           // the two inversions side by side
        store_affine_be<M>(out + ((size_t)i + (r ? len : 0)) * 64, xyzz30_to_xyzz<M>(xyzz30_load_lazy<M>(&L.qd[q].tbl[1 + r])));
}

template <class M>
__device__ __forceinline__ void store_affine_be(uint8_t* dst, const XYZZ<M>& p) {
    if (xyzz_is_inf<M>(p)) {
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4* q = reinterpret_cast<uint4*>(dst);
        q[0] = z; q[1] = z; q[2] = z; q[3] = z;
        return;
    }
    Fe<M> inv = fe_inv_safegcd<M>(p.zzz);
    Affine<M> a = xyzz_to_affine_with_inv<M>(p, inv);
    Fe<M> one = fe_zero<M>();
    one.v[0] = 1;
    Fe<M> x = fe_mul_call<M>(a.x, one), y = fe_mul_call<M>(a.y, one);
    store_be256(dst, x.v);
    store_be256(dst + 32, y.v);
}

// MAC part of Server::mix (Server.hpp:1281-1318): out[i] = A0[i] + v^i * A1[i], out[i+len] = A0[i] - v^i * A1[i], v = w^(N/len);
// 64-byte big-endian affine points in and out.  One lane per i.
template <class C>
__global__ void __launch_bounds__(64)
k_mac_mix(const uint8_t* __restrict__ a0, const uint8_t* __restrict__ a1, uint32_t len, const uint32_t* __restrict__ tws,
          uint32_t tw_step, uint8_t* __restrict__ out, const uint8_t* __restrict__ b0, const uint8_t* __restrict__ b1,
          uint8_t* __restrict__ out_b) {
    using M = typename C::Fp;
    if (blockIdx.y) { a0 = b0; a1 = b1; out = out_b; }          // the second array pair of a mix (see k_mac_mix_quad)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    uint32_t sc[8];
    const uint4* q = reinterpret_cast<const uint4*>(tws + (size_t)i * tw_step * 8);
    uint4 a = q[0], b = q[1];
    sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    XYZZ<M> hi = load_affine_be_lazy<M>(a1 + (size_t)i * 64);
    XYZZ<M> tm;
    mac30_scalar_mul<C>(&tm, &hi, sc);
    XYZZ<M> sum = load_affine_be_lazy<M>(a0 + (size_t)i * 64);
    XYZZ<M> dif = sum;
    xyzz30_add_mem<M>(&sum, &tm, 0, 0, nullptr);
    xyzz30_add_mem<M>(&dif, &tm, 1, 0, nullptr);
    store_affine_be<M>(out + (size_t)i * 64, xyzz30_to_xyzz<M>(xyzz30_load_lazy<M>(&sum)));
    store_affine_be<M>(out + ((size_t)i + len) * 64, xyzz30_to_xyzz<M>(xyzz30_load_lazy<M>(&dif)));
}

// XYZZ work array -> 64-byte big-endian affine MACs (infinity = 64 zero bytes, main.go:224-230)
template <class C>
__global__ void __launch_bounds__(64)
k_mac_finish(const XYZZ<typename C::Fp>* __restrict__ work, uint32_t n, uint8_t* __restrict__ out) {
    using M = typename C::Fp;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<M> p = load_xyzz<M>(work + i);
    p = xyzz30_to_xyzz<M>(xyzz30_load_lazy<M>(&p));                              // the work array's lazy memory form
    uint8_t* dst = out + (size_t)i * 64;
    if (xyzz_is_inf<M>(p)) {
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4* q = reinterpret_cast<uint4*>(dst);
        q[0] = z; q[1] = z; q[2] = z; q[3] = z;
        return;
    }
    Fe<M> inv = fe_inv_safegcd<M>(p.zzz);
    Affine<M> a = xyzz_to_affine_with_inv<M>(p, inv);
    Fe<M> one = fe_zero<M>();
    one.v[0] = 1;
    Fe<M> x = fe_mul_call<M>(a.x, one), y = fe_mul_call<M>(a.y, one);
    store_be256(dst, x.v);
    store_be256(dst + 32, y.v);
}

}  // namespace porla
