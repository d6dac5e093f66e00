// The ICC encode kernel of the reduced-radix form with the two residues of a symbol treated as two PLANES and two butterfly
// stages per LDS round trip -- the same network (Server::CRebuild_Cached, porla/Server/Server.hpp:1548-1687 X part, :1691-1830
// Y part; the twiddle product t = vi * X[k+m2] of :1649-1655), the same tiles and passes, the same arithmetic per residue as
// the round-2 kernel that held both residues side by side, so the outputs are bit for bit the same.  What changes is how a block holds its tile:
//
//   * Z/LCM = Z/p_icc x Z/q: nothing couples the residue mod p_icc and the residue mod q of a symbol before the finish step.
//     A block therefore runs the ns stages of its tile for the p_icc plane first and for the q plane second, through ONE LDS
//     region of ICC_TILE_ELEMS slots x 40 bytes (1 024: 40 KiB; round 3-4: 512 slots, 20 KiB) (the side-by-side form of round 2: 80-byte slots, 40 KiB).  Four 256-lane blocks fit a CU (round 3-4: eight 128-lane ones) with half
//     the symbols per lane in flight of the side-by-side form: the product of fe30.hip.h is one dependent chain of
//     multiply-adds, a lone wave issues it at half rate, and a SIMD needs two READY waves to keep its multiplier busy -- with
//     four resident waves that each spend a fifth of their life in an LDS round trip or at a barrier it often has one.
//   * RADIX 4 in registers: a lane reads the four symbols {m, m + 2^d, m + 2^(d+1), m + 3 2^d} of a tile once, runs stage d on
//     the pairs (0,1), (2,3) and stage d+1 on (0,2), (1,3) -- three twiddles: v_d^j, v_(d+1)^j and v_(d+1)^(j + m2) = table entry
//     + N/2, as icc.hip.h:k_icc_stages<2> does through HBM -- and writes them back once: half the LDS traffic, half the barriers.
//     An odd stage count ends with a radix-2 round of two butterflies per lane.
//   * The FIRST round of a plane takes its symbols straight from memory (at d = 0 a unit is four neighbouring rows) and the LAST
//     round leaves them in registers: from there to the work plane (between the passes) or, in the last pass, into the finish
//     step -- ns = 8 stages are three LDS round trips per plane instead of nine.  The finish step needs both residues: the lane
//     parks A mod p_icc (32 bytes) in `aligned` -- or, when the caller does not want it, in the symbol's own place in the p_icc
//     work plane -- and reads it back when the q plane is done (36 registers it would otherwise hold across the q plane).
//   * The work set between the passes is two plane arrays of 9 words per symbol (72 bytes per symbol as before, each plane read
//     once by the phase that needs it); the twiddle table likewise (40-byte slots per plane).
//   * blockIdx -> tile is dealt so that an XCD (blocks b, b + 8, ...) owns a contiguous range of tiles: the column tiles that
//     share the 128-byte lines of a row run on one L2.
#pragma once
#include "icc30.hip.h"

namespace porla {

constexpr int ICC30_PLANE_WORDS = 9;      // a plane's symbol in the work set between the passes
constexpr int ICC30_PSLOT_WORDS = 10;     // LDS / twiddle slot of a plane (8-byte accesses, stride 10 words: conflict-free)
#ifndef PORLA_ICC_SPLIT_THREADS
#define PORLA_ICC_SPLIT_THREADS (PORLA_ICC_TILE / 4)
#endif
constexpr int ICC30_SPLIT_THREADS = PORLA_ICC_SPLIT_THREADS;      // a quarter of the tile: one radix-4 unit per lane and round

template <class M>
__device__ __forceinline__ F30<M> icc30_ld_pslot(const uint32_t* s) {          // 8-byte aligned
    const uint2* q = reinterpret_cast<const uint2*>(s);
    const uint2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    F30<M> r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y; r.v[4] = c.x; r.v[5] = c.y; r.v[6] = d.x; r.v[7] = d.y; r.v[8] = e.x;
    return r;
}
template <class M>
__device__ __forceinline__ void icc30_st_pslot(uint32_t* s, const F30<M>& r) {
    uint2* q = reinterpret_cast<uint2*>(s);
    q[0] = make_uint2(r.v[0], r.v[1]); q[1] = make_uint2(r.v[2], r.v[3]); q[2] = make_uint2(r.v[4], r.v[5]);
    q[3] = make_uint2(r.v[6], r.v[7]); q[4] = make_uint2(r.v[8], 0u);
}
// a plane's symbol in the work set: 9 words, 4-byte aligned
struct IccW9 { uint32_t w[9]; };
template <class M>
__device__ __forceinline__ F30<M> icc30_ld_work(const uint32_t* s) {
    const IccW9 t = *reinterpret_cast<const IccW9*>(s);
    F30<M> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = t.w[i];
    return r;
}
template <class M>
__device__ __forceinline__ void icc30_st_work(uint32_t* s, const F30<M>& r) {
    IccW9 t;
#pragma unroll
    for (int i = 0; i < 9; i++) t.w[i] = r.v[i];
    *reinterpret_cast<IccW9*>(s) = t;
}

// the plane tables of the twiddles: entry e = w^e in the 2^270 form, 40-byte slots (entry 0 = the Montgomery unit)
template <class Q>
__global__ void __launch_bounds__(256)
k_icc_twiddles30_planes(const IccElem<Q>* __restrict__ tw, uint32_t n, uint32_t* __restrict__ twp, uint32_t* __restrict__ twq) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const IccElem30<Q> t = icc30_from_elem<Q>(ld_elem<Q>(tw + e));
    icc30_st_pslot<IccFp>(twp + (size_t)e * ICC30_PSLOT_WORDS, t.p);
    icc30_st_pslot<Q>(twq + (size_t)e * ICC30_PSLOT_WORDS, t.q);
}

// one butterfly on one plane: (a, b) <- (a + w b, a - w b + 2 p); the product brings b below p + 2^248 whatever it was
template <class M>
__device__ __forceinline__ void icc30_bfly(F30<M>& a, F30<M>& b, const F30<M>& w) {
    const F30<M> t = icc30_mul<M>(w, b);
    b = icc30_sub<M, 2>(a, t);
    a = icc30_add<M>(a, t);
}
// a butterfly whose twiddle is w^0 and whose lower operand is known to be small: (a, b) <- (a + b, a - b + K p), b <= (K - 1) p
// + 2^240.  Stage 1 of the first pass: operands straight from the load step -- products of it (K = 2: b < p + 2^248) or RAW 256-bit
// chunks (K = 7: b < 2^256 < 6 p + 2^240 for all three moduli, the smallest being the BN254 group order at 0.189 x 2^256)
template <class M, int K>
__device__ __forceinline__ void icc30_bfly_plain(F30<M>& a, F30<M>& b) {
    static_assert(K <= 4 || (unsigned __int128)(K - 1) * ((((unsigned __int128)M::P[7]) << 32) | M::P[6]) >= ((unsigned __int128)1 << 64),
                  "K = 7 is for raw 256-bit operands: (K - 1) p must exceed 2^256 (K <= 4: the caller bounds b by (K - 1) p itself)");
    const F30<M> t = b;
    b = icc30_sub<M, K>(a, t);
    a = icc30_add<M>(a, t);
}

struct IccTile {           // what a block knows about its tile (the same for both planes)
    uint32_t n, ncols, elems, cc_log, lo_bits, lo, row_base, c0;
    int s0, ns;
    int raw;               // first pass without an init scaling: the chunks enter the network as they are (see icc30_fetch)
};

// a tile symbol from where the pass finds it.  FIRST: the raw 32-byte chunk x.  The stream is made of PLAIN residues and nothing
// in the network needs them reduced -- a product with a twiddle brings any operand below 2^263 under p + 2^248, sums and
// differences stay unreduced anyway -- so without an init scaling (the X part: T.raw) the 256-bit chunk enters as it is: nine
// normal limbs, value below 2^256 (the first stage's difference then adds 7 p instead of 2 p, and a symbol is bounded by
// 8.3 p + 2 p per later stage: 66.3 p < 2^263 after 30 stages -- limb 8 below 2^23, the column bound tools/check_fe30_bounds.py proves -- like the
// 63 p of the scaled form).  With an init scaling (the Y part) the load step is the product x * (wt 2^270) / 2^270 = x wt.
// (icc30.hip.h:icc30_load_raw multiplies by the Montgomery unit in the unscaled case: two products per symbol that only reduce.)
// Not FIRST: the plane's work array.
template <class M, bool FIRST>
__device__ __forceinline__ F30<M> icc30_fetch(const IccTile& T, uint32_t e, const uint8_t* __restrict__ raw, const F30<M>& K,
                                              const uint32_t* __restrict__ work) {
    const uint32_t mid = e >> T.cc_log, col = e & ((1u << T.cc_log) - 1u);
    const size_t gi = (size_t)(T.row_base + (mid << T.lo_bits)) * T.ncols + T.c0 + col;
    if (FIRST) {
        const Fe<IccFp> x = ld_fe<IccFp>(reinterpret_cast<const uint32_t*>(raw + 32 * gi));
        const F30<M> u = f30_unpack<M>(x.v);
        return T.raw ? u : icc30_mul<M>(u, K);
    }
    return icc30_ld_work<M>(work + gi * ICC30_PLANE_WORDS);
}

// One round of a plane: stages s = s0 + d and (RADIX4) s + 1 on the lane's symbols.  GLOBAL_IN: the pass's first round takes its
// symbols straight from memory (d = 0: the four rows of a unit are neighbours), later rounds from the LDS tile; KEEP: the pass's
// last round leaves them in res[] / slot[] instead of the tile.  RADIX4: one unit m0 + {0,1,2,3} 2^d per lane; else two
// butterflies (m0, m0 + 2^d) per lane.
template <class M, bool FIRST, bool GLOBAL_IN, bool RADIX4>
__device__ __forceinline__ void icc30_round(uint32_t* lds, const IccTile& T, int d, bool keep, const uint8_t* __restrict__ raw,
                                            const F30<M>& K, const uint32_t* __restrict__ work, const uint32_t* __restrict__ tw,
                                            F30<M> (&res)[4], uint32_t (&slot)[4]) {
    const uint32_t tid = threadIdx.x;
    const uint32_t Cc = 1u << T.cc_log;
    const int s = T.s0 + d;
    if (RADIX4) {
        const uint32_t col = tid & (Cc - 1), qq = tid >> T.cc_log;
        if (tid < (T.elems >> 2) && T.c0 + col < T.ncols) {
            const uint32_t low = qq & ((1u << d) - 1u);
            const uint32_t m0 = ((qq >> d) << (d + 2)) | low;
            uint32_t e[4];
            F30<M> a[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                e[i] = ((m0 | ((uint32_t)i << d)) << T.cc_log) + col;
                a[i] = GLOBAL_IN ? icc30_fetch<M, FIRST>(T, e[i], raw, K, work) : icc30_ld_pslot<M>(lds + (size_t)e[i] * ICC30_PSLOT_WORDS);
            }
            const uint32_t j = (low << T.lo_bits) + T.lo;                        // row index mod m2 of stage s
            if (FIRST && GLOBAL_IN) {
                // stage 1 of the encode: every twiddle is w^0 and both operands come straight from the load step
                if (T.raw) { icc30_bfly_plain<M, 7>(a[0], a[1]); icc30_bfly_plain<M, 7>(a[2], a[3]); }
                else { icc30_bfly_plain<M, 2>(a[0], a[1]); icc30_bfly_plain<M, 2>(a[2], a[3]); }
            } else {
                const F30<M> w1 = icc30_ld_pslot<M>(tw + (size_t)j * (T.n >> (s - 1)) * ICC30_PSLOT_WORDS);
                icc30_bfly<M>(a[0], a[1], w1);
                icc30_bfly<M>(a[2], a[3], w1);
            }
            // stage s + 1: m2' = 2^s; pair (0, 2): j' = j; pair (1, 3): j' = j + 2^(s-1) -> table entry + N / 2
            const size_t i2 = (size_t)j * (T.n >> s);
            if (FIRST && GLOBAL_IN) {
                // the encode's first round: j = 0 for EVERY unit, the twiddle of pair (0, 2) is w^0 -- a product with the
                // Montgomery unit would only reduce a[2].  Whole waves skip it: a[2] = a2 + a3 is below 2^257 (raw chunks; a short
                // reduction brings it under 2 p, K = 3) or below 2 (p + 2^248) (scaled chunks, K = 4); the outputs stay under
                // 2^257 + 3 p, inside the bound icc30_fetch states
                if (T.raw) { a[2] = icc30_reduce_top<M>(a[2]); icc30_bfly_plain<M, 3>(a[0], a[2]); }
                else icc30_bfly_plain<M, 4>(a[0], a[2]);
            } else {
                const F30<M> w2 = icc30_ld_pslot<M>(tw + i2 * ICC30_PSLOT_WORDS);
                icc30_bfly<M>(a[0], a[2], w2);
            }
            {
                const F30<M> w3 = icc30_ld_pslot<M>(tw + (i2 + (T.n >> 1)) * ICC30_PSLOT_WORDS);
                icc30_bfly<M>(a[1], a[3], w3);
            }
            if (keep) {
#pragma unroll
                for (int i = 0; i < 4; i++) { res[i] = a[i]; slot[i] = e[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) icc30_st_pslot<M>(lds + (size_t)e[i] * ICC30_PSLOT_WORDS, a[i]);
            }
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t bf = tid + h * ICC30_SPLIT_THREADS;
            const uint32_t col = bf & (Cc - 1), qq = bf >> T.cc_log;
            if (bf < (T.elems >> 1) && T.c0 + col < T.ncols) {
                const uint32_t low = qq & ((1u << d) - 1u);
                const uint32_t m0 = ((qq >> d) << (d + 1)) | low;
                const uint32_t e0 = (m0 << T.cc_log) + col, e1 = ((m0 | (1u << d)) << T.cc_log) + col;
                F30<M> a = GLOBAL_IN ? icc30_fetch<M, FIRST>(T, e0, raw, K, work) : icc30_ld_pslot<M>(lds + (size_t)e0 * ICC30_PSLOT_WORDS);
                F30<M> b = GLOBAL_IN ? icc30_fetch<M, FIRST>(T, e1, raw, K, work) : icc30_ld_pslot<M>(lds + (size_t)e1 * ICC30_PSLOT_WORDS);
                if (FIRST && GLOBAL_IN) {
                    if (T.raw) icc30_bfly_plain<M, 7>(a, b);
                    else icc30_bfly_plain<M, 2>(a, b);
                } else {
                    const uint32_t j = (low << T.lo_bits) + T.lo;
                    const F30<M> w = icc30_ld_pslot<M>(tw + (size_t)j * (T.n >> (s - 1)) * ICC30_PSLOT_WORDS);
                    icc30_bfly<M>(a, b, w);
                }
                if (keep) {
                    res[2 * h] = a; slot[2 * h] = e0;
                    res[2 * h + 1] = b; slot[2 * h + 1] = e1;
                } else {
                    icc30_st_pslot<M>(lds + (size_t)e0 * ICC30_PSLOT_WORDS, a);
                    icc30_st_pslot<M>(lds + (size_t)e1 * ICC30_PSLOT_WORDS, b);
                }
            }
        }
    }
}

// ns stages of ONE plane of the block's tile: radix-4 rounds, then a radix-2 round when ns is odd.  Out: the lane's (up to) four
// symbols of the last round in res[], their tile slots in slot[] (0xffffffff: none).  The caller syncs before the LDS region is
// used again (the last round only reads it).
template <class M, bool FIRST>
__device__ __forceinline__ void icc30_plane(uint32_t* lds, const IccTile& T, const uint8_t* __restrict__ raw, const F30<M>& K,
                                            const uint32_t* __restrict__ work, const uint32_t* __restrict__ tw,
                                            F30<M> (&res)[4], uint32_t (&slot)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) slot[i] = 0xffffffffu;
    if (T.ns == 1) {
        icc30_round<M, FIRST, true, false>(lds, T, 0, true, raw, K, work, tw, res, slot);
        return;
    }
    icc30_round<M, FIRST, true, true>(lds, T, 0, T.ns == 2, raw, K, work, tw, res, slot);
    int d = 2;
    while (d + 1 < T.ns) {
        __syncthreads();
        icc30_round<M, false, false, true>(lds, T, d, d + 2 == T.ns, raw, K, work, tw, res, slot);
        d += 2;
    }
    if (d < T.ns) {
        __syncthreads();
        icc30_round<M, false, false, false>(lds, T, d, true, raw, K, work, tw, res, slot);
    }
}

// LDS-fused stages s0 .. s0+ns-1 of a tile of 2^ns rows x 2^cc_log columns (<= ICC_TILE_ELEMS symbols), both planes; grid = tiles
// XY (last pass only): the outputs of BOTH parts from this one network.  The network is linear over Z/LCM and the Y part is the X
// part's network on chunks scaled by wt (Server.hpp:1494, :1512-1522, then the same stages :1691-1830), so Y_k = wt X_k mod LCM,
// residue by residue: one product per plane and symbol on the last round's registers, then the same finish step with `out_y`'s
// pointers -- instead of a second run of every pass.  use_wt must be 0 (the network is the X part's); wt256 is the scaling.
// A mod p_icc of the Y part waits for the q plane in out_y.al or, when the caller does not want it, in park_y (32 B per symbol).
template <class Q, bool FIRST, bool LAST, bool XY = false>
__global__ void __launch_bounds__(ICC30_SPLIT_THREADS) __attribute__((amdgpu_waves_per_eu(XY ? 3 : 4, 4)))
k_icc_split30(uint32_t* __restrict__ work_p, uint32_t* __restrict__ work_q, const uint32_t* __restrict__ twp,
              const uint32_t* __restrict__ twq, uint32_t n, uint32_t ncols, int s0, int ns, int cc_log,
              const uint8_t* __restrict__ raw, IccElem<Q> wt256, int use_wt, IccOut out, IccOut out_y, uint32_t* __restrict__ park_y) {
    static_assert(LAST || !XY, "the second part's outputs are derived in the last pass");
    __shared__ uint2 lds2[ICC_TILE_ELEMS * ICC30_PSLOT_WORDS / 2];
    uint32_t* lds = reinterpret_cast<uint32_t*>(lds2);
    IccTile T;
    T.n = n; T.ncols = ncols; T.s0 = s0; T.ns = ns; T.cc_log = (uint32_t)cc_log;
    T.elems = (1u << ns) << cc_log;
    T.raw = FIRST && !use_wt;
    T.lo_bits = (uint32_t)(s0 - 1);
    const uint32_t Cc = 1u << cc_log;
    const uint32_t col_tiles = (ncols + Cc - 1) >> cc_log;
    // blocks b, b + 8, ... share an XCD (observed round-robin dispatch; speed only): XCD x takes tiles [x G / 8, (x + 1) G / 8)
    uint32_t tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t ct = tile % col_tiles;
    tile /= col_tiles;
    T.lo = tile & ((1u << T.lo_bits) - 1u);
    const uint32_t hi = tile >> T.lo_bits;
    T.row_base = (hi << (T.lo_bits + ns)) + T.lo;
    T.c0 = ct << cc_log;

    // planes the requested outputs need: the values mod p_icc (al) only the p_icc plane, the values mod q (qres: the MAC side's
    // network matrix) only the q plane; the alignment scalars and the values mod LCM both
    const bool y_p = XY && (out_y.x || out_y.al || out_y.sc), y_q = XY && (out_y.x || out_y.sc || out_y.qres);
    const bool x_q = out.x || out.sc || out.qres;
    const bool need_p = !LAST || out.x || out.al || out.sc || y_p;
    const bool need_q = !LAST || x_q || y_q;
    uint32_t slot[4];
    if (need_p) {
        F30<IccFp> rp[4];
        F30<IccFp> K;
        if (FIRST && use_wt) K = icc30_mul<IccFp>(f30_unpack<IccFp>(wt256.p.v), f30_const<IccFp>(Icc30Const<IccFp>::C284));
        else K = F30<IccFp>{};
        icc30_plane<IccFp, FIRST>(lds, T, raw, K, work_p, twp, rp, slot);
        F30<IccFp> Ky = F30<IccFp>{};
        if (XY) Ky = icc30_mul<IccFp>(f30_unpack<IccFp>(wt256.p.v), f30_const<IccFp>(Icc30Const<IccFp>::C284));   // wt in the 2^270 form
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (slot[i] != 0xffffffffu) {
                const uint32_t mid = slot[i] >> cc_log, col = slot[i] & (Cc - 1);
                const size_t gi = (size_t)(T.row_base + (mid << T.lo_bits)) * ncols + T.c0 + col;
                if (LAST) {
                    // A mod p_icc leaves the registers here: to `aligned` when the caller wants it, and -- for the q plane's finish
                    // step, by the same lane -- to the symbol's own place in the p_icc work plane, which this block has finished reading
                    if (out.x || out.al || out.sc) {
                        const Fe<IccFp> P = icc30_finish_p(rp[i]);
                        if (out.al) st_fe<IccFp>(reinterpret_cast<uint32_t*>(out.al + 32 * gi), P);
                        if ((out.x || out.sc) && !out.al) {
                            uint32_t* d = work_p + gi * ICC30_PLANE_WORDS;
#pragma unroll
                            for (int k = 0; k < 8; k++) d[k] = P.v[k];
                        }
                    }
                    if (y_p) {
                        const Fe<IccFp> P = icc30_finish_p(icc30_mul<IccFp>(rp[i], Ky));      // (wt X) mod p_icc
                        if (out_y.al) st_fe<IccFp>(reinterpret_cast<uint32_t*>(out_y.al + 32 * gi), P);
                        if ((out_y.x || out_y.sc) && !out_y.al) st_fe<IccFp>(park_y + gi * 8, P);
                    }
                } else {
                    icc30_st_work<IccFp>(work_p + gi * ICC30_PLANE_WORDS, rp[i]);
                }
            }
        }
        __syncthreads();                                 // every lane has read its last round's symbols: the region is free
    }
    if (need_q) {
        F30<Q> rq[4];
        F30<Q> K;
        if (FIRST && use_wt) K = icc30_mul<Q>(f30_unpack<Q>(wt256.q.v), f30_const<Q>(Icc30Const<Q>::C284));
        else K = F30<Q>{};
        icc30_plane<Q, FIRST>(lds, T, raw, K, work_q, twq, rq, slot);
        F30<Q> Ky = F30<Q>{};
        if (XY) Ky = icc30_mul<Q>(f30_unpack<Q>(wt256.q.v), f30_const<Q>(Icc30Const<Q>::C284));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (slot[i] != 0xffffffffu) {
                const uint32_t mid = slot[i] >> cc_log, col = slot[i] & (Cc - 1);
                const size_t gi = (size_t)(T.row_base + (mid << T.lo_bits)) * ncols + T.c0 + col;
                if (LAST) {
                    if (x_q) {
                        Fe<IccFp> P = fe_zero<IccFp>();
                        if (out.x || out.sc) {               // what this lane stored after the p_icc plane
                            const uint32_t* src = out.al ? reinterpret_cast<const uint32_t*>(out.al + 32 * gi) : work_p + gi * ICC30_PLANE_WORDS;
#pragma unroll
                            for (int k = 0; k < 8; k++) P.v[k] = src[k];
                        }
                        icc30_finish_q<Q>(P, rq[i], gi, out);
                    }
                    if (y_q) {
                        Fe<IccFp> P = fe_zero<IccFp>();
                        if (out_y.x || out_y.sc) {
                            const uint32_t* src = out_y.al ? reinterpret_cast<const uint32_t*>(out_y.al + 32 * gi) : park_y + gi * 8;
#pragma unroll
                            for (int k = 0; k < 8; k++) P.v[k] = src[k];
                        }
                        icc30_finish_q<Q>(P, icc30_mul<Q>(rq[i], Ky), gi, out_y);               // (wt X) mod q
                    }
                } else {
                    icc30_st_work<Q>(work_q + gi * ICC30_PLANE_WORDS, rq[i]);
                }
            }
        }
    }
}

}  // namespace porla
