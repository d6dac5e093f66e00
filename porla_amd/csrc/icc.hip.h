// ICC (incrementally constructible code) encode on gfx950: the radix-2 butterfly network that the reference
// open-codes on NTL big integers in Server::CRebuild_Cached (porla/Server/Server.hpp:1548-1687 X part, :1691-1830
// Y part, init scaling :1494,1512-1522) and the scalar part of Server::align_MAC (Server.hpp:531-541 KZG,
// :495-504 IPA).  Constants: porla/Utils/utils.h:27-43.
//
// WHAT is computed (bit-exact): for s = 1..log2 N: m = 2^s, m2 = m/2, v = w^(N/m2) mod p_icc; for j < m2, k = j (step m):
//   t = v^j * X[k+m2];  X[k] = (X[k] + t) mod LCM;  X[k+m2] = (X[k] - t) mod LCM,   LCM = p_icc * q,
// element-wise over 128-wide rows, natural order in, no permutation, w = GENERATOR^((p_icc-1)/(2N)) (order N).
//
// HOW (MI355X): Z/LCM = Z/p_icc x Z/q (both prime), so every element is carried as a pair of 256-bit Montgomery
// residues -- two 256-bit products per butterfly instead of one 512-bit product plus a 768->512-bit reduction -- and
// recombined once at the end with p_icc = 207*2^248 + 1 (a shift-and-add CRT).  The encode is bound by the integer
// multiplier, not by HBM: a block keeps a tile of 512 symbols (2^ns rows x 2^(9-ns) columns) in LDS through ns <= 8 stages, so
// 15 stages are two passes over the working set (32 B in + 2 x 64..72 B between the passes + the outputs per symbol), against
// ~1 000 vector instructions per symbol and pass.  The first pass converts the raw 32-byte chunks on the way in, the last writes
// the CRT / alignment outputs; the N-entry twiddle table stays L2-resident; lanes run along the columns of a row, so global
// accesses are coalesced.  The encode kernel is k_icc_split30 of icc30_split.hip.h (9 x 30-bit limbs, radix 2^270, sparse reduction
// for p_icc, one residue plane at a time); the element-wise kernels of this file (k_icc_load, k_icc_finish) serve HAdd.
// No MFMA: there is no contraction.
#pragma once
#include "fe.hip.h"

namespace porla {

struct IccFp {  // p_icc = 207 * 2^248 + 1 (utils.h:31-32)
    static constexpr uint32_t P[8]  = {0x00000001u, 0, 0, 0, 0, 0, 0, 0xcf000000u};
    static constexpr uint32_t INV   = 0xffffffffu;
    static constexpr uint32_t R1[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu,
                                       0xffffffffu, 0xffffffffu, 0xffffffffu, 0x30ffffffu};
    static constexpr uint32_t R2[8] = {0xe1150631u, 0xfb0d9a96u, 0x845418bbu, 0xec366a5bu,
                                       0x115062efu, 0xb0d9a96eu, 0x45418bbfu, 0x6266a5b8u};
    static constexpr int SPARE_BITS = 0;
    static constexpr bool PSEUDO_MERSENNE = false;
    static constexpr uint32_t FOLD = 0;
};
struct IccBn254Fr {  // q = BN254 group order (utils.h:36), with the CRT constant p_icc^-1 mod q
    static constexpr uint32_t P[8]  = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                       0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV   = 0xefffffffu;
    static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                       0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr int SPARE_BITS = 2;
    static constexpr uint32_t PINV[8] = {0xf7cceac7u, 0x74a3c74fu, 0x32b19079u, 0xa1c013fcu,
                                         0xdaf96adbu, 0xe53d9858u, 0x7d59bd98u, 0x2d305e1eu};  // p_icc^-1 mod q (plain)
    static constexpr int MAX_Q_IN = 5;   // floor((2^256-1)/q): subtractions to reduce a raw 256-bit chunk
    static constexpr int MAX_Q_P  = 4;   // floor((p_icc-1)/q)
    static constexpr bool PSEUDO_MERSENNE = false;
    static constexpr uint32_t FOLD = 0;
};
struct IccSecp256k1Fn {  // q = secp256k1 group order (utils.h:27)
    static constexpr uint32_t P[8]  = {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u,
                                       0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    static constexpr uint32_t INV   = 0x5588b13fu;
    static constexpr uint32_t R1[8] = {0x2fc9bebfu, 0x402da173u, 0x50b75fc4u, 0x45512319u, 0x00000001u, 0, 0, 0};
    static constexpr uint32_t R2[8] = {0x67d7d140u, 0x896cf214u, 0x0e7cf878u, 0x741496c2u,
                                       0x5bcd07c6u, 0xe697f5e4u, 0x81c69bc5u, 0x9d671cd5u};
    static constexpr int SPARE_BITS = 0;
    static constexpr uint32_t PINV[8] = {0x62768be0u, 0x61996758u, 0x398bec83u, 0x15831c63u,
                                         0x74f03b2fu, 0x43030579u, 0xeb08b927u, 0x8d224d74u};
    static constexpr int MAX_Q_IN = 1;
    static constexpr int MAX_Q_P  = 0;
    static constexpr bool PSEUDO_MERSENNE = false;
    static constexpr uint32_t FOLD = 0;
};
// GENERATOR (utils.h:29-30), plain little-endian limbs
struct IccGen {
    static constexpr uint32_t G[8] = {0x1ea0a8b6u, 0x3bd639dau, 0xc05b565bu, 0x8daf5cecu,
                                      0x693fe88eu, 0x7b4b58b0u, 0x5de0999du, 0x001559f5u};
};

template <class Q>
struct IccElem {  // one code symbol: residues mod p_icc and mod q, Montgomery form; 64 B
    Fe<IccFp> p;
    Fe<Q> q;
};

template <class M>
__device__ __forceinline__ Fe<M> ld_fe(const uint32_t* s) {
    const uint4* q = reinterpret_cast<const uint4*>(s);
    uint4 a = q[0], b = q[1];
    Fe<M> f;
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    return f;
}
template <class M>
__device__ __forceinline__ void st_fe(uint32_t* d, const Fe<M>& f) {
    uint4* q = reinterpret_cast<uint4*>(d);
    q[0] = make_uint4(f.v[0], f.v[1], f.v[2], f.v[3]);
    q[1] = make_uint4(f.v[4], f.v[5], f.v[6], f.v[7]);
}
template <class Q>
__device__ __forceinline__ IccElem<Q> ld_elem(const IccElem<Q>* p) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(p);
    IccElem<Q> e;
    e.p = ld_fe<IccFp>(s);
    e.q = ld_fe<Q>(s + 8);
    return e;
}
template <class Q>
__device__ __forceinline__ void st_elem(IccElem<Q>* p, const IccElem<Q>& e) {
    uint32_t* d = reinterpret_cast<uint32_t*>(p);
    st_fe<IccFp>(d, e.p);
    st_fe<Q>(d + 8, e.q);
}

// raw 256-bit little-endian chunk (utils.h:353-364) -> residue pair, optionally times wt (the Y part)
template <class Q>
__device__ __forceinline__ IccElem<Q> icc_load_elem(const uint8_t* src, const IccElem<Q>& wt, int use_wt) {
    Fe<IccFp> xp = ld_fe<IccFp>(reinterpret_cast<const uint32_t*>(src));
    Fe<Q> xq;
#pragma unroll
    for (int k = 0; k < 8; k++) xq.v[k] = xp.v[k];
    fe_reduce_plain<IccFp>(xp.v, 1);
    fe_reduce_plain<Q>(xq.v, Q::MAX_Q_IN);
    IccElem<Q> e;
    e.p = fe_to_mont<IccFp>(xp);
    e.q = fe_to_mont<Q>(xq);
    if (use_wt) {
        e.p = fe_mul<IccFp>(e.p, wt.p);
        e.q = fe_mul<Q>(e.q, wt.q);
    }
    return e;
}
template <class Q>
__global__ void k_icc_load(const uint8_t* __restrict__ in, IccElem<Q>* __restrict__ work, size_t total,
                           IccElem<Q> wt, int use_wt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    st_elem<Q>(work + i, icc_load_elem<Q>(in + 32 * i, wt, use_wt));
}

// tw[e] = w^e for e in [0, n): residue pair of the INTEGER (w^e mod p_icc), as the reference multiplies by the
// integer vi_ZZ (Server.hpp:1651).  wpow[i] = w^(2^i) in Montgomery form mod p_icc.
template <class Q>
__global__ void k_icc_twiddles(IccElem<Q>* __restrict__ tw, uint32_t n, const Fe<IccFp>* __restrict__ wpow, int logn) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    Fe<IccFp> acc = fe_one<IccFp>();
    for (int i = 0; i < logn; i++) {
        if ((e >> i) & 1) acc = fe_mul<IccFp>(acc, wpow[i]);
    }
    Fe<IccFp> plain = fe_from_mont<IccFp>(acc);
    Fe<Q> tq;
#pragma unroll
    for (int k = 0; k < 8; k++) tq.v[k] = plain.v[k];
    fe_reduce_plain<Q>(tq.v, Q::MAX_Q_P + 1);
    IccElem<Q> t;
    t.p = acc;
    t.q = fe_to_mont<Q>(tq);
    st_elem<Q>(tw + e, t);
}

// value in [0, LCM) as 64 bytes little-endian from the residue pair: P = A mod p_icc (plain), pq_m = P mod q (Montgomery)
template <class Q>
__device__ __forceinline__ void icc_store_lcm_pt(const Fe<IccFp>& P, const Fe<Q>& t, uint8_t* dst);
template <class Q>
__device__ __forceinline__ void icc_store_lcm(const IccElem<Q>& e, const Fe<IccFp>& P, const Fe<Q>& pq_m, uint8_t* dst) {
        // A = P + p_icc * t,  t = (a_q - P) * p_icc^-1 mod q;  p_icc * t = t + (207 t << 248)
    Fe<Q> pinv;
#pragma unroll
    for (int k = 0; k < 8; k++) pinv.v[k] = Q::PINV[k];
    icc_store_lcm_pt<Q>(P, fe_mul<Q>(fe_sub<Q>(e.q, pq_m), pinv), dst);   // Montgomery(d) * plain -> plain product
}
// A = P + t + (207 t << 248) as 64 bytes little-endian (P = A mod p_icc, t = the CRT coefficient, both plain)
template <class Q>
__device__ __forceinline__ void icc_store_lcm_pt(const Fe<IccFp>& P, const Fe<Q>& t, uint8_t* dst) {
    uint32_t u[9];
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t x = (uint64_t)t.v[k] * 207u + carry;
        u[k] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
    u[8] = carry;
    uint32_t A[16];
#pragma unroll
    for (int k = 0; k < 16; k++) A[k] = 0;
    // A = P + t
    uint32_t c2 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) A[k] = adc32(P.v[k], t.v[k], c2);
    A[8] = c2;
    // A += u << 248   (248 = 7*32 + 24)
    uint32_t c3 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint32_t lo = u[k] << 24;
        uint32_t hi = (k > 0) ? (u[k - 1] >> 8) : 0;
        A[7 + k] = adc32(A[7 + k], lo | hi, c3);
    }
    // top word of (u << 248): u[8] >> 8 lands in A[16] which must be zero because A < LCM < 2^512
    uint4* o = reinterpret_cast<uint4*>(dst);
    o[0] = make_uint4(A[0], A[1], A[2], A[3]);
    o[1] = make_uint4(A[4], A[5], A[6], A[7]);
    o[2] = make_uint4(A[8], A[9], A[10], A[11]);
    o[3] = make_uint4(A[12], A[13], A[14], A[15]);
}

// residue pair -> (a) value in [0, LCM) as 64-byte LE, (b) value mod p_icc as 32-byte LE,
// (c) alignment scalar c = (A mod p_icc - A) mod q (Server.hpp:535-538) as 32 bytes BE (or LE limbs),
// (d) A mod q as 32 bytes BE (the coefficient the MAC side sees, mac_fft.hip)
struct IccOut {   // output pointers of the finish step (each may be null)
    uint8_t* x;      // 64 B LE, value mod LCM
    uint8_t* al;     // 32 B LE, value mod p_icc
    uint8_t* sc;     // 32 B alignment scalar (BE, or LE limbs with scalar_le)
    uint8_t* qres;   // 32 B BE, value mod q
    int scalar_le;
};
template <class Q>
__device__ __forceinline__ void icc_finish_elem(const IccElem<Q>& e, size_t i, const IccOut& o) {
    uint8_t* const x_out = o.x; uint8_t* const al_out = o.al; uint8_t* const sc_out = o.sc; uint8_t* const qres_out = o.qres;
    const int scalar_le = o.scalar_le;
    if (qres_out) {
        Fe<Q> a = fe_from_mont<Q>(e.q);
        uint4* q4 = reinterpret_cast<uint4*>(qres_out + 32 * i);
        q4[0] = make_uint4(__builtin_bswap32(a.v[7]), __builtin_bswap32(a.v[6]), __builtin_bswap32(a.v[5]), __builtin_bswap32(a.v[4]));
        q4[1] = make_uint4(__builtin_bswap32(a.v[3]), __builtin_bswap32(a.v[2]), __builtin_bswap32(a.v[1]), __builtin_bswap32(a.v[0]));
        if (!x_out && !al_out && !sc_out) return;
    }
    Fe<IccFp> P = fe_from_mont<IccFp>(e.p);      // A mod p_icc, plain
    if (al_out) st_fe<IccFp>(reinterpret_cast<uint32_t*>(al_out + 32 * i), P);
    Fe<Q> pq;                                     // (A mod p_icc) mod q
#pragma unroll
    for (int k = 0; k < 8; k++) pq.v[k] = P.v[k];
    fe_reduce_plain<Q>(pq.v, Q::MAX_Q_P + 1);
    Fe<Q> pq_m = fe_to_mont<Q>(pq);
    if (sc_out) {
        Fe<Q> c = fe_from_mont<Q>(fe_sub<Q>(pq_m, e.q));
        uint32_t* d = reinterpret_cast<uint32_t*>(sc_out + 32 * i);
        if (scalar_le) {
            st_fe<Q>(d, c);
        } else {
            uint4* q4 = reinterpret_cast<uint4*>(d);
            q4[0] = make_uint4(__builtin_bswap32(c.v[7]), __builtin_bswap32(c.v[6]), __builtin_bswap32(c.v[5]), __builtin_bswap32(c.v[4]));
            q4[1] = make_uint4(__builtin_bswap32(c.v[3]), __builtin_bswap32(c.v[2]), __builtin_bswap32(c.v[1]), __builtin_bswap32(c.v[0]));
        }
    }
    if (x_out) icc_store_lcm<Q>(e, P, pq_m, x_out + 64 * i);
}
template <class Q>
__global__ void k_icc_finish(const IccElem<Q>* __restrict__ work, size_t total, IccOut o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    icc_finish_elem<Q>(ld_elem<Q>(work + i), i, o);
}

// The LDS-fused encode (icc30_split.hip.h) gives a block a TILE: 2^ns rows {row_base + mid * 2^(s0-1)} (all values of the ns
// row-index bits its stages pair up; the lower s0-1 bits `lo` and the upper bits `hi` are fixed per tile) x 2^cc_log columns =
// ICC_TILE_ELEMS symbols.  The tile is read from HBM once, goes through ns butterfly stages in LDS and is written back once: 15 stages cost
// 2 passes over the working set instead of 8.
#ifndef PORLA_ICC_TILE
#define PORLA_ICC_TILE 1024        // round 5: 1 024 symbols (4 .. 8 columns of a row per tile: 144 .. 288 contiguous bytes per plane and row) -- same-box
#endif                              // A/B against 512 and 2 048: 0.583 / 0.566 / 0.579 ms per 2^15 x 128 encode (profiles/r05_o_icc_tile_ab.txt)
constexpr int ICC_TILE_ELEMS = PORLA_ICC_TILE;
constexpr int ICC_TILE_LOG = PORLA_ICC_TILE == 2048 ? 11 : (PORLA_ICC_TILE == 1024 ? 10 : (PORLA_ICC_TILE == 512 ? 9 : 8));

// value of LIMBS (17..24) 32-bit limbs reduced into Montgomery form mod M: Horner over 256-bit digits with R = 2^256
template <class M, int LIMBS>
PORLA_HD Fe<M> icc_reduce_wide(const uint32_t* a) {
    Fe<M> r2, d2, d1, d0;
#pragma unroll
    for (int k = 0; k < 8; k++) { r2.v[k] = M::R2[k]; d0.v[k] = a[k]; d1.v[k] = a[8 + k]; d2.v[k] = (k < LIMBS - 16) ? a[16 + k] : 0; }
    // Montgomery product with R2 takes any 256-bit operand: x -> x * R mod M
    Fe<M> acc = fe_mul<M>(d2, r2);
    acc = fe_add<M>(fe_mul<M>(acc, r2), fe_mul<M>(d1, r2));
    acc = fe_add<M>(fe_mul<M>(acc, r2), fe_mul<M>(d0, r2));
    return acc;
}

// 512-bit little-endian value (< LCM, a stored code symbol) -> residue pair: Horner over the two 256-bit halves, R = 2^256
template <class M>
__device__ __forceinline__ Fe<M> icc_reduce512(const uint32_t a[16]) {
    Fe<M> r2, d1, d0;
#pragma unroll
    for (int k = 0; k < 8; k++) { r2.v[k] = M::R2[k]; d0.v[k] = a[k]; d1.v[k] = a[8 + k]; }
    // Montgomery product with R2 takes any 256-bit operand: x -> x * R mod M
    Fe<M> acc = fe_mul<M>(d1, r2);
    return fe_add<M>(fe_mul<M>(acc, r2), fe_mul<M>(d0, r2));
}

// rows_in = N x N identity (32-byte LE chunks): entry (i, i) = 1
static __global__ void k_icc_identity(uint8_t* __restrict__ rows, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[((size_t)i * n + i) * 32] = 1;
}

}  // namespace porla
