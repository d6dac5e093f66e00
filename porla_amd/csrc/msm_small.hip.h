// Single-launch MSM for the sizes the reference actually issues and a margin (n <= 32 768 pairs; the audit's n_points <= 3 200,
// porla/Server/Server.hpp:585-587, 838-848, 900-901; the client's 176- and 1 408-point calls, porla/Client/Client.hpp:664-669,
// 756-787) on both curves.
//
// The general path (msm.hip.h) is built for throughput: ~20 dependent launches, a sort through global memory and a 15-level
// tree over mostly empty buckets cost 0.3-0.5 ms whatever the input.  Here ONE kernel does everything and what is left is
// a chain of ~15 dependent group additions:
//
//   grid = up to 256 blocks of 256 lanes (one block per CU, one wave per SIMD: dependent chains run at the lone-wave latency);
//   192 up to 4 096 pairs and 2 x 96 for a pair of MSMs (msm_impl.hip.h): as fast, and the rest of the chip stays free.
//   Every block first ORs all n scalars (<= 128 KB, L2 resident) to learn their bit length -- no host round trip: the audit's
//   abs(int32) coefficients (utils.h:271-275) then need 9 windows of 4 bits instead of 32 -- and derives the shape from it:
//   scalars longer than 128 bits are split with the curve endomorphism (k = k1 + lambda k2, glv.hip.h; 2n sub-scalars of
//   <= 128 bits), so the host fold is <= 130 doublings instead of 255.
//   block (w, s) owns window w and slice s of the sub-scalars:
//     1. signed c-bit digits of its slice for its window (carries walked from window 0), counting-sorted by bucket in LDS
//     2. T = 256 / buckets lanes share a bucket: each accumulates every T-th entry with the mixed addition (points are
//        converted from the 64-byte wire format on the fly: the 2 extra products per addition are cheaper than a launch)
//     3. the T partial sums of a bucket are folded, then the bucket reduction runs as the bit-sliced tree of msm.hip.h
//        (S and M_k per node) -- all in LDS, four lanes per addition (ec30.hip.h:xyzz30_add_quad)
//     4. the block's c sums (S, M_0 .. M_(c-2)) go to global memory; the LAST block of a window to arrive (one atomic per
//        block) folds the slices' sums and writes fin[w][*] straight into pinned host memory, next to the shape (W, c)
//   host: the same Horner fold over single bits as the general path (host_fold64.hpp).
//
// Same arithmetic as the general path (digits, endomorphism split, mixed / full additions are the very same functions), so the
// result is the same group element and the marshalled 64 bytes are bit-exact.
#pragma once
#include "msm.hip.h"

namespace porla {

constexpr uint32_t SMALL_MAX_N = 32768;             // (the reference stops at 3 200; up to here one launch still beats the general path's ~25)
constexpr int SMALL_THREADS = 256;
constexpr int SMALL_BLOCKS = 256;
constexpr int SMALL_MAX_C = 8;                       // <= 128 buckets per window
constexpr int SMALL_MAX_B = 1 << (SMALL_MAX_C - 1);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "msm_small.hip.h sizes its per-block state (about 105 KB of static LDS) for gfx950's 160 KB of LDS per CU; there is no smaller-LDS variant"
#endif
constexpr uint32_t SMALL_MAX_SUB = 8192;             // sub-scalars ONE block (a slice of a window) can hold: LDS, 13-bit local index
constexpr int SMALL_MAX_S = 32;                      // slices per window: S * c <= 256 sums fit the LDS fold
constexpr int SMALL_DONE_SLOT = 255;                 // counters[0 .. W): arrivals per window; [255]: finished windows
constexpr uint32_t SMALL_HDR_WORDS = 32;             // pinned header in front of fin: [0] = sequence, [1] = W, [2] = c, [3] = glv

// the single-launch commitment of fixed_base.hip.h (k_fb_commit_small): rows per launch, slices per row
constexpr int FB_SMALL_MAX_ROWS = 64;            // 64 rows x 8 slices = 512 blocks, two per compute unit (96 rows: 0.19 ms, the batch kernels 0.22)
constexpr int FB_SMALL_MAX_SLICES = 8;

struct SmallCfg {
    int glv;          // 1: every scalar split in two sub-scalars
    int L;            // bits a sub-scalar can have
    int c, W, S;      // window bits, windows, slices per window
    uint32_t n_sub;
};

// used_bits: bit length of the OR of the raw 256-bit scalars.  order_bits: bit length of the group order.
// c_flags: window bits override in the low byte (0 = automatic), 0x100 = never split (porla_gpu_set_msm_glv(0))
template <class C>
__host__ __device__ inline SmallCfg small_cfg(uint32_t n, int used_bits, int c_flags, int blocks) {
    SmallCfg g;
    const int c_override = c_flags & 0xff;
    // below 2^128 the scalar itself is short (and certainly below the group order: no reduction, no split)
    g.glv = (used_bits > 128 && !(c_flags & 0x100)) ? 1 : 0;
    // unsplit: a scalar that may reach the group order is reduced first and then has at most SCALAR_BITS bits
    g.L = g.glv ? C::Glv::BITS : (used_bits < 1 ? 1 : (used_bits >= 250 ? C::SCALAR_BITS : used_bits));
    g.n_sub = g.glv ? 2 * n : n;
    // Window width: every block runs log2(256) = 8 dependent additions whatever c is (fold of the lanes of a bucket + bucket
    // tree), then ceil(log2 S) more to fold the slices, S = blocks / W: MORE windows mean FEWER dependent additions, until the
    // slices get so large that a lane has several entries to accumulate one after the other.  The estimate (half-microseconds,
    // fitted to tools/small_stamps.hip): 18 per accumulated entry after the first (gather + conversion + mixed addition),
    // 7 per tree level, 4 (split) or 2 per scalar a lane recodes; a bucket's load is taken at its mean + 3 sigma.
    int best_c = 0;
    float best_cost = 1e30f;
    // (every block evaluates this before it can start: the loop is unrolled so that the divisions by c are by constants, and the
    // two by run-time values are float divisions -- exact here: blocks <= 256, n <= 32 768, S <= 32 and the quotients are floored)
#pragma unroll
    for (int c = 1; c <= SMALL_MAX_C; c++) {
        if (c_override >= 1 && c_override <= SMALL_MAX_C && c != c_override) continue;
        const int W = (g.L + c) / c;
        if (W > blocks || W >= SMALL_DONE_SLOT) continue;
        uint32_t S = (uint32_t)((float)blocks / (float)W);
        const uint32_t by_size = (g.n_sub + 7) / 8;      // a slice of fewer than 8 sub-scalars is not worth a fold level
        if (S > by_size) S = by_size;
        if (S > (uint32_t)SMALL_MAX_S) S = SMALL_MAX_S;
        if (S < 1) S = 1;
        if ((g.n_sub + S - 1) / S > SMALL_MAX_SUB) continue;      // a slice must fit the block's LDS
        const float Bf = (float)(1u << (c > 1 ? c - 1 : 0));
        const float per_bucket = (float)g.n_sub / (float)S * ((1.0f - 1.0f / (float)(1u << c)) / Bf);
        const float lanes = 256.0f / Bf;
        float per_lane = (per_bucket + 3.0f * __builtin_sqrtf(per_bucket)) / lanes;
        int chain = (int)per_lane;
        if ((float)chain < per_lane) chain++;
        int lv = 0;
        while ((1u << lv) < S) lv++;
        const int recode = (int)(((uint32_t)((float)n / (float)S) + 255u) / 256u);
        const float cost = 18.0f * (float)(chain > 1 ? chain - 1 : 0) + 7.0f * (float)(8 + lv) + (g.glv ? 4.0f : 2.0f) * (float)recode +
                           0.01f * (float)W;
        if (cost < best_cost) { best_cost = cost; best_c = c; g.c = c; g.W = W; g.S = (int)S; }
    }
    if (best_c == 0) {                                   // no admissible shape (a forced width with more windows than blocks): c = 8 always fits (W <= 32)
        g.c = SMALL_MAX_C; g.W = (g.L + 1 + g.c - 1) / g.c;
        g.S = blocks / g.W < 1 ? 1 : (blocks / g.W > SMALL_MAX_S ? SMALL_MAX_S : blocks / g.W);
    }
    return g;
}

// `tasks` independent additions on the block's quads; get(t, &pa, &pb, &out) names the operands of task t (memory form)
template <class M, class F>
__device__ __forceinline__ void small_quad_adds(uint32_t tasks, F get) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t t0 = 0; t0 < tasks; t0 += SMALL_THREADS / 4) {
        uint32_t t = t0 + (threadIdx.x >> 2);
        const bool live = t < tasks;
        if (!live) t = tasks - 1;
        const XYZZ<M>*pa, *pb;
        XYZZ<M>* out;
        get(t, pa, pb, out);
        xyzz30_add_quad<M>(pa, pb, out, false, live, lane);
    }
}

// phase stamps of one block for tools/small_stamps.hip (never compiled into the library)
#ifdef PORLA_SMALL_STAMPS
#define SMALL_STAMP(k) do { if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define SMALL_STAMP_ARG , unsigned long long* __restrict__ stamps
#else
#define SMALL_STAMP(k) do { } while (0)
#define SMALL_STAMP_ARG
#endif

template <class C>
__global__ void __launch_bounds__(SMALL_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_small_msm(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points, uint32_t n, int c_override,
            XYZZ<typename C::Fp>* __restrict__ part, uint32_t* __restrict__ counters, uint32_t* __restrict__ hdr,
            XYZZ<typename C::Fp>* __restrict__ fin, uint32_t seq, const uint8_t* __restrict__ points_b,
            uint32_t pair_stride_bytes SMALL_STAMP_ARG) {
    using M = typename C::Fp;
    // gridDim.y == 2: the audit's PAIR of MSMs -- the same scalars over two point sets (combined_MAC and combined_align,
    // porla/Server/Server.hpp:842-848 / :900-901) -- in one launch: set blockIdx.y takes its own points, partial sums, arrival
    // counters and pinned result region (header + window sums at pair_stride_bytes); each set shapes itself for gridDim.x blocks
    if (blockIdx.y) {
        points = points_b;
        part += (size_t)gridDim.x * SMALL_MAX_C;
        counters += 256;
        hdr = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(hdr) + pair_stride_bytes);
        fin = reinterpret_cast<XYZZ<M>*>(reinterpret_cast<uint8_t*>(fin) + pair_stride_bytes);
    }
    __shared__ uint32_t ent[SMALL_MAX_SUB];                    // sorted entries: sub-scalar index | sign << 31
    __shared__ XYZZ<M> pts[SMALL_THREADS];                     // first the unsorted digits (uint32 view), then the lane sums
    __shared__ XYZZ<M> bk[SMALL_MAX_B];                        // bucket sums
    __shared__ XYZZ<M> slev[SMALL_MAX_B];                      // S levels of the tree: B/2 + B/4 + ... + 1
    __shared__ XYZZ<M> mlev[SMALL_MAX_B / 2 + 2];              // M slots, two ping-pong halves
    __shared__ uint32_t hist[SMALL_MAX_B], cursor[SMALL_MAX_B];
    __shared__ uint32_t orw[8], pat[8];
    __shared__ uint32_t last_flag;
    const uint32_t tid = threadIdx.x;
    uint32_t* raw = reinterpret_cast<uint32_t*>(pts);          // SMALL_THREADS * 32 words >= SMALL_MAX_SUB

    SMALL_STAMP(0);
    // ---- 0. bit length of the scalars (every block for itself) -- unless the caller knows a bound: bits 16..24 of c_override (the
    // audit's coefficients are abs(int32), expanded to 32-byte scalars by k_audit_gather: 32 bits, no scan -- ~10 us of a ~100 us kernel)
    const int bits_hint = (c_override >> 16) & 0x1ff;
    if (tid < 8) orw[tid] = 0;
    if (tid < SMALL_MAX_B) hist[tid] = 0;
    __syncthreads();
    if (bits_hint == 0) {
        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // four scalars per lane and trip, loaded before the first is used: the loop is a chain of L2 round trips otherwise
        // (13 trips of ~0.6 us at 3 200 scalars)
        for (uint32_t i = tid; i < n; i += 4 * SMALL_THREADS) {
            uint32_t t[4][8];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = i + (uint32_t)u * SMALL_THREADS;
                load_be256(t[u], scalars + (size_t)(j < n ? j : i) * 32);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int k = 0; k < 8; k++) acc[k] |= t[u][k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t v = acc[k];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m, 64);
            if ((tid & 63) == 0 && v) atomicOr(&orw[k], v);
        }
    }
    __syncthreads();
    int used = bits_hint;
    if (bits_hint == 0) {
#pragma unroll
        for (int k = 7; k >= 0; k--)
            if (used == 0 && orw[k]) used = 32 * k + (32 - __clz(orw[k]));
    }
    const SmallCfg g = small_cfg<C>(n, used, c_override, (int)gridDim.x);
    if (blockIdx.x >= (uint32_t)(g.W * g.S)) return;
    const uint32_t w = blockIdx.x / g.S, s = blockIdx.x % g.S;
    const int c = g.c;
    const uint32_t mask = (1u << c) - 1;
    const uint32_t Bfull = 1u << (c - 1);
    // the top window holds only the bits left over (plus a carry): magnitudes 1 .. 2^(top-1), never negative -- its blocks run
    // with that many buckets (and more lanes per bucket) instead of leaving most lanes idle behind a few crowded buckets
    const int cw = (w + 1 == (uint32_t)g.W) ? g.L + 1 - c * (g.W - 1) : c;      // 1 .. c
    const uint32_t B = 1u << (cw - 1);
    const bool reduce = used >= 250;                           // a raw scalar may reach the group order: fr.SetBytes reduces (main.go:127)
    const uint32_t SUBS = g.glv ? 2u : 1u;
    const uint32_t i0 = (uint32_t)((uint64_t)s * n / g.S), i1 = (uint32_t)((uint64_t)(s + 1) * n / g.S);   // the slice's scalars

    const uint32_t wc = w * (uint32_t)c;                          // first bit of the window
    if (tid < 8) {                                             // bits p of limb tid with p mod c == c - 1
        uint32_t v = 0;
        for (uint32_t b = (uint32_t)c - 1 - (32 * tid) % (uint32_t)c; b < 32; b += (uint32_t)c) v |= 1u << b;
        pat[tid] = v;
    }
    __syncthreads();
    SMALL_STAMP(1);
    // ---- 1. digits of window w for the slice, unsorted into raw[], histogram by bucket
    for (uint32_t i = i0 + tid; i < i1; i += SMALL_THREADS) {
        uint32_t k[8];
        load_be256(k, scalars + (size_t)i * 32);
        if (reduce) {
            for (int q = 0; q < C::MAX_Q; q++) {
                uint32_t d[8];
                uint32_t br = 0;
#pragma unroll
                for (int q2 = 0; q2 < 8; q2++) d[q2] = sbb32(k[q2], C::ORDER[q2], br);
                if (br) break;
#pragma unroll
                for (int q2 = 0; q2 < 8; q2++) k[q2] = d[q2];
            }
        }
        uint32_t sub[2][8];
        uint32_t sneg[2] = {0, 0};
        if (g.glv) {
            uint32_t m1[4], m2[4];
            bool n1, n2;
            glv_split<typename C::Glv>(k, m1, n1, m2, n2);
#pragma unroll
            for (int q = 0; q < 4; q++) { sub[0][q] = m1[q]; sub[1][q] = m2[q]; sub[0][4 + q] = 0; sub[1][4 + q] = 0; }
            sneg[0] = n1 ? 1u : 0u; sneg[1] = n2 ? 1u : 0u;
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) { sub[0][q] = k[q]; sub[1][q] = 0; }
        }
#pragma unroll
        for (uint32_t e = 0; e < 2; e++) {
            if (e >= SUBS) break;
            // signed digit of window w.  The recoding "digit > B: subtract 2^c, carry" is the unique representation with digits in
            // (-B, B], so the carry into window w is 1 exactly when the bits below the window exceed B * (1 + 2^c + ... ) = the
            // pattern with bit c-1 of every lower window set (pat[]): one masked compare instead of a walk over w windows.
            bool gt = false, eq = true;
#pragma unroll
            for (int q = 7; q >= 0; q--) {
                const uint32_t below = wc > 32u * q ? wc - 32u * q : 0u;       // bits of limb q below the window
                const uint32_t m = below >= 32u ? 0xffffffffu : ((1u << below) - 1u);
                const uint32_t a = sub[e][q] & m, b = pat[q] & m;
                gt = eq ? (a > b) : gt;
                eq = eq && (a == b);
            }
            uint32_t rawd = gt ? 1u : 0u;
            if (wc < 256u) {
                const uint32_t limb = wc >> 5, sh = wc & 31u;
                uint32_t a = 0, b = 0;
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) {
                    a = (q == limb) ? sub[e][q] : a;
                    b = (q == limb + 1) ? sub[e][q] : b;
                }
                rawd += (uint32_t)((((uint64_t)b << 32) | a) >> sh) & mask;
            }
            uint32_t mag = rawd, dneg = 0;
            if (rawd > Bfull) { mag = (1u << c) - rawd; dneg = 1; }
            const uint32_t j = (i - i0) * SUBS + e;
            uint32_t packed = 0xffffffffu;
            if (mag) {
                packed = j | ((dneg ^ sneg[e]) << 13) | ((mag - 1) << 14);
                atomicAdd(&hist[mag - 1], 1u);
            }
            raw[j] = packed;
        }
    }
    __syncthreads();
    SMALL_STAMP(2);
    // exclusive scan of the B <= 128 counters (Hillis-Steele in LDS)
    if (tid < SMALL_MAX_B) cursor[tid] = tid < B ? hist[tid] : 0;
    __syncthreads();
    for (uint32_t d = 1; d < B; d <<= 1) {
        uint32_t v = 0;
        if (tid < B && tid >= d) v = cursor[tid - d];
        __syncthreads();
        if (tid < B) cursor[tid] += v;
        __syncthreads();
    }
    if (tid < B) cursor[tid] -= hist[tid];                     // exclusive: start of the bucket's run
    __syncthreads();
    const uint32_t T = SMALL_THREADS / B;                      // lanes per bucket (a power of two, 2 .. 256)
    const uint32_t my_b = tid / T, my_t = tid % T;
    const uint32_t my_start = cursor[my_b], my_cnt = hist[my_b];
    __syncthreads();
    for (uint32_t j = tid; j < (i1 - i0) * SUBS; j += SMALL_THREADS) {
        const uint32_t p = raw[j];
        if (p != 0xffffffffu) {
            const uint32_t pos = atomicAdd(&cursor[p >> 14], 1u);
            ent[pos] = (i0 * SUBS + (p & 0x1fffu)) | (((p >> 13) & 1u) << 31);
        }
    }
    __syncthreads();

    SMALL_STAMP(3);
    // ---- 2. accumulate: lane (bucket, t) takes entries t, t + T, ... of its bucket
    {
        XYZZ30<M> acc;
        acc.inf = true;
        bool flip = false;
        // the next entry's point is in flight while this one is converted and added (a gather is an L2 / HBM round trip)
        uint32_t en = 0;
        Affine<M> nx;
        if (my_t < my_cnt) {
            en = ent[my_start + my_t];
            const uint32_t i = g.glv ? (en & 0x7fffffffu) >> 1 : (en & 0x7fffffffu);
            load_be256(nx.x.v, points + (size_t)i * 64);
            load_be256(nx.y.v, points + (size_t)i * 64 + 32);
        }
        for (uint32_t e = my_t; e < my_cnt; e += T) {
            const uint32_t cur = en;
            Affine<M> a = nx;
            if (e + T < my_cnt) {
                en = ent[my_start + e + T];
                const uint32_t i = g.glv ? (en & 0x7fffffffu) >> 1 : (en & 0x7fffffffu);
                load_be256(nx.x.v, points + (size_t)i * 64);
                load_be256(nx.y.v, points + (size_t)i * 64 + 32);
            }
            fe_reduce_plain<M>(a.x.v, 6);                      // G1Affine.Unmarshal: SetBytes reduces (main.go:130)
            fe_reduce_plain<M>(a.y.v, 6);
            if (aff_is_inf<M>(a)) continue;
            // (the accumulator alternates sign with every addition, ec30.hip.h:xyzz30_madd_flip; folded into the digit's sign)
            a = aff_neg_if<M>(a, xyzz30_flip_neg<M>((cur >> 31) != 0, flip));   // on the plain residue: -y = p - y in any form
            // into the 2^270 form the reduced-radix addition computes in: one product by 2^540 mod p per coordinate (the special-form
            // field keeps plain residues: nothing to do); phi(P) = (beta x, y) for the second sub-scalar
            F30<M> ax = f30_from_fe<M>(a.x), ay = f30_from_fe<M>(a.y);
            if constexpr (!M::PSEUDO_MERSENNE) {
                ax = f30_mul<M>(ax, f30_const<M>(M::RR_30));
                ay = f30_mul<M>(ay, f30_const<M>(M::RR_30));
            }
            if (g.glv && (cur & 1u)) ax = f30_mul<M>(ax, f30_const<M>(C::Glv::BETA_30));
            xyzz30_madd_flip<M>(acc, flip, ax, ay);
        }
        xyzz30_flip_finish<M>(acc, flip);
        xyzz30_store_lazy<M>(&pts[tid], acc);
    }
    __syncthreads();

    SMALL_STAMP(4);
    // ---- 3a. fold the T lane sums of every bucket (the last step writes the compact bucket array)
    for (uint32_t h = 1; h < T; h <<= 1) {
        const bool last = (h << 1) == T;
        small_quad_adds<M>(SMALL_THREADS / (2 * h), [&](uint32_t t, const XYZZ<M>*& pa, const XYZZ<M>*& pb, XYZZ<M>*& out) {
            pa = &pts[t * 2 * h]; pb = pa + h;
            out = last ? &bk[t] : &pts[t * 2 * h];
        });
        __syncthreads();
    }
    SMALL_STAMP(5);
    // ---- 3b. bucket reduction: the bit-sliced tree of msm.hip.h on one window's B buckets, in LDS
    const uint32_t nlev = (uint32_t)(cw - 1);
    auto s_level = [&](uint32_t l) { return slev + (B - (B >> l)); };     // B/2 + ... + B/2^l entries before level l
    auto m_half = [&](uint32_t h) { return mlev + (h & 1u) * (SMALL_MAX_B / 4 + 1); };
    for (uint32_t l = 0; l < nlev; l++) {
        const uint32_t nl = B >> (l + 1);
        const XYZZ<M>* sp = l ? s_level(l - 1) : bk;
        const XYZZ<M>* sp2 = l >= 2 ? s_level(l - 2) : bk;
        const XYZZ<M>* mp = m_half(l + 1);
        XYZZ<M>* so = s_level(l);
        XYZZ<M>* mo = m_half(l);
        small_quad_adds<M>((l + 1) * nl, [&](uint32_t t, const XYZZ<M>*& pa, const XYZZ<M>*& pb, XYZZ<M>*& out) {
            const uint32_t sl = t / nl, i = t % nl;
            if (sl == l) { pa = sp + 2 * i; pb = pa + 1; out = so + i; }
            else if (sl + 1 == l) { pa = sp2 + 4 * i + 1; pb = pa + 2; out = mo + sl * nl + i; }
            else { pa = mp + sl * 2 * nl + 2 * i; pb = pa + 1; out = mo + sl * nl + i; }
        });
        __syncthreads();
    }
    SMALL_STAMP(6);
    // the block's c sums: S, M_0 .. M_(c-2)  (the last one the tree produces is the alias S^(nlev-2)[1] -- or bucket 1 when there
    // is one level; the top window's sums beyond its own width are infinity = zero words)
    XYZZ<M>* mine = part + (size_t)blockIdx.x * SMALL_MAX_C;
    if (tid < (uint32_t)c * 8) {
        const uint32_t kk = tid >> 3, q = tid & 7;             // 8 lanes copy the 128 bytes of sum kk
        const XYZZ<M>* src = nullptr;
        if (kk == 0) src = nlev ? s_level(nlev - 1) : bk;
        else if (kk > nlev) src = nullptr;
        else if (kk == nlev) src = (nlev >= 2 ? s_level(nlev - 2) : bk) + 1;
        else src = m_half(nlev - 1) + (kk - 1);
        reinterpret_cast<uint4*>(mine + kk)[q] = src ? reinterpret_cast<const uint4*>(src)[q] : make_uint4(0, 0, 0, 0);
    }
    // ---- 4. the last block of the window folds the slices
    SMALL_STAMP(7);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const uint32_t old = atomicAdd(&counters[w], 1u);
        last_flag = (old == (uint32_t)g.S - 1) ? 1u : 0u;
        if (last_flag) counters[w] = 0;                        // ready for the next launch on this workspace
    }
    __syncthreads();
    if (!last_flag) return;
    __threadfence();
    SMALL_STAMP(8);
    // the S * c sums of the window come into LDS in one pass (slice sl, sum k at pts[sl * c + k]; S <= SMALL_MAX_S)
    {
        const uint4* src = reinterpret_cast<const uint4*>(part + (size_t)w * g.S * SMALL_MAX_C);
        uint4* dst = reinterpret_cast<uint4*>(pts);
        const uint32_t total = (uint32_t)g.S * (uint32_t)c * 8;
        for (uint32_t x = tid; x < total; x += SMALL_THREADS) {
            const uint32_t e = x >> 3, q = x & 7;
            dst[x] = src[(size_t)((e / (uint32_t)c) * SMALL_MAX_C + e % (uint32_t)c) * 8 + q];
        }
    }
    __syncthreads();
    for (uint32_t cnt = (uint32_t)g.S; cnt > 1;) {
        const uint32_t half = (cnt + 1) / 2, pairs = cnt - half;
        small_quad_adds<M>(pairs * (uint32_t)c, [&](uint32_t t, const XYZZ<M>*& pa, const XYZZ<M>*& pb, XYZZ<M>*& out) {
            const uint32_t sl = t / (uint32_t)c, kk = t % (uint32_t)c;
            pa = &pts[sl * c + kk]; pb = &pts[(sl + half) * c + kk];
            out = &pts[sl * c + kk];
        });
        __syncthreads();
        cnt = half;
    }
    if (tid < (uint32_t)c) Node<C>::store_final(fin + (size_t)w * c + tid, Node<C>::load(&pts[tid]));
    SMALL_STAMP(9);
    // the last window to finish publishes the shape and the sequence number the host polls for
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const uint32_t old = atomicAdd(&counters[SMALL_DONE_SLOT], 1u);
        if (old == (uint32_t)g.W - 1) {
            counters[SMALL_DONE_SLOT] = 0;
            hdr[1] = (uint32_t)g.W; hdr[2] = (uint32_t)c; hdr[3] = (uint32_t)g.glv;
            __threadfence_system();
            __atomic_store_n(&hdr[0], seq, __ATOMIC_RELEASE);
        }
    }
}

}  // namespace porla
