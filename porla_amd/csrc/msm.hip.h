// Pippenger bucket multi-scalar multiplication for gfx950 (MI355X), templated on the curve.
//
// Replaces the reference's MSM providers behind their own boundaries:
//   BN254     compute_multi_exp -> G1Affine.MultiExp          porla/main.go:118-138
//   secp256k1 secp256k1_ecmult_multi_var (pippenger_batch)     porla/Utils/secp256k1_lib/ecmult_impl.h:814-860, :646-720
// The reference algorithms (gnark: signed c-bit windows + extended-Jacobian buckets, one goroutine
// per window; libsecp256k1: GLV + wNAF + buckets, single thread) fix WHAT is computed
// (sum (s_i mod order) * P_i); the decomposition below is designed for 256 CUs:
//
//   k_points_to_mont    64-B big-endian affine -> Montgomery limbs (one coalesced pass, 64 B in / 64 B out)
//   k_digits_partition  32-B big-endian scalar -> reduce mod order -> signed c-bit digits, grouped per 4096-scalar tile
//                       by bucket partition in LDS (no global atomics)
//   k_partition_sort    block per (window, partition): LDS counting sort -> counts/starts per bucket + point indices
//                       grouped by bucket (coalesced reads of the tile runs)
//   k_size_hist/scan/order  work items (<= CHUNK entries of one bucket) counting-sorted BY SIZE, largest first, so the 64
//                       lanes of a wave run the same trip count; a heavy bucket becomes many items (skew-proof)
//   k_bucket_sum30      ONE THREAD PER WORK ITEM: walks its index list, gathers 64-B points (L2 / Infinity-Cache
//                       resident: 2^20 points = 64 MiB) and accumulates with the 8M+2S mixed add in the reduced-radix
//                       field form of fe30.hip.h / ec30.hip.h (k_bucket_sum: the same on 8 x 32-bit limbs, kept for curves
//                       without that form)
//   k_bucket_combine    wave per multi-item bucket: folds that bucket's item sums (no-op for uniform scalars)
//   k_tree_level(_quad), k_tree_tail   per window sum_b (b+1)*B_b as a bit-sliced tree: S (plain sum) and M_k (sum of the
//                       buckets with index bit k set) per node, log2(B) levels of independent additions; small levels run
//                       each addition on the four lanes of a quad, the last levels in one block per window
//   host                one Horner pass over the W*c single-bit terms (host_fold64.hpp), affine normalisation, marshal
//
// Traffic per pair (c = 17, W = 15): 96 B input + 64 B converted point + 2 * W * 4 B sort items + W * 4 B index + W gathers
// of a 64-B point (the 64 MiB point set lives in the Infinity Cache; PMC: 3.1 GB of fabric requests per 2^20-pair launch).
// The accumulation is VALU (integer multiply) bound, see DESIGN.md s4.
#pragma once
#include "ec.hip.h"
#include "ec30.hip.h"
#include "glv.hip.h"
#include <type_traits>

namespace porla {

struct Bn254G1 {
    using Fp = Bn254Fp;
    // group order r (scalar field), little-endian 32-bit limbs
    static constexpr uint32_t ORDER[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                          0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr int SCALAR_BITS = 254;
    static constexpr int MAX_Q = 5;  // floor((2^256-1)/r)
    using Glv = GlvBn254;
    // 254-bit scalars fill 16 windows of 16 bits exactly; the GLV split would halve the windows but double the entries per
    // bucket and the gathered point set (measured: 3.31 ms vs 2.85 ms at 2^20) -- off by default, on with porla_gpu_set_msm_glv(1)
    static constexpr bool GLV_DEFAULT = false;
    static constexpr size_t GLV_BELOW = (size_t)1 << 17;   // ... but on up to this many pairs (msm_impl.hip.h:msm_use_glv)
    // bucket accumulation in the reduced-radix field form (fe30.hip.h / ec30.hip.h): 1.35-1.6x the field-product rate
    static constexpr bool F30_BUCKETS = true;
    // ... and the sums stay in that form ("lazy" memory form of ec30.hip.h: unreduced residues, X <= 5p < 2^256) through the
    // combine and tree kernels
    static constexpr bool F30_LAZY = true;
    static constexpr int BUCKET_SUM_WAVES = 4;   // waves per SIMD k_bucket_sum30 is compiled for
    static constexpr int FB_COMMIT_WAVES = 3;    // ... and k_fb_commit (fixed_base.hip.h: 161-169 registers by build; 168 is the most three waves leave each)
    static constexpr int MACQ_WAVES = 4;         // the quad-lane MAC kernels: 128 registers, they run beside the commitments of a CRebuild (mac_fft.hip.h)
};
struct Secp256k1G {
    using Fp = Secp256k1Fp;
    static constexpr uint32_t ORDER[8] = {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u,
                                          0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    static constexpr int SCALAR_BITS = 256;
    static constexpr int MAX_Q = 1;
    using Glv = GlvSecp256k1;
    // 256-bit scalars need a carry-only 17th window at c = 16 (one bucket with half of all entries); the GLV split (which
    // the reference's secp256k1 path also applies, ecmult_impl.h:621-634) gives 8 windows of 17 bits instead
    static constexpr bool GLV_DEFAULT = true;
    static constexpr size_t GLV_BELOW = 0;
    static constexpr bool F30_BUCKETS = true;    // special-form product on 30-bit limbs: 194 against 133 G products/s
    static constexpr bool F30_LAZY = true;       // memory form: canonical residues (5p > 2^256: an unreduced X does not fit 32 bytes)
    static constexpr int BUCKET_SUM_WAVES = 3;   // 148 registers; held to 120 for four waves it measures the same (698-700 Mmul/s both, round 5)
    static constexpr int FB_COMMIT_WAVES = 2;    // k_fb_commit: ~205 registers
    static constexpr int MACQ_WAVES = 2;         // no combined CRebuild stage on this curve: the quad-lane MAC kernels take the registers the fold wants (no spills)
};

constexpr uint32_t KEY_NONE = 0xffffffffu;

// ------------------------------------------------------------------------------------------------
// 32 big-endian bytes -> 8 little-endian 32-bit limbs (two 16-byte loads)
__device__ __forceinline__ void load_be256(uint32_t t[8], const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 hi = q[0], lo = q[1];
    t[7] = __builtin_bswap32(hi.x); t[6] = __builtin_bswap32(hi.y);
    t[5] = __builtin_bswap32(hi.z); t[4] = __builtin_bswap32(hi.w);
    t[3] = __builtin_bswap32(lo.x); t[2] = __builtin_bswap32(lo.y);
    t[1] = __builtin_bswap32(lo.z); t[0] = __builtin_bswap32(lo.w);
}
__device__ __forceinline__ void store_be256(uint8_t* p, const uint32_t t[8]) {
    uint4 hi, lo;
    hi.x = __builtin_bswap32(t[7]); hi.y = __builtin_bswap32(t[6]);
    hi.z = __builtin_bswap32(t[5]); hi.w = __builtin_bswap32(t[4]);
    lo.x = __builtin_bswap32(t[3]); lo.y = __builtin_bswap32(t[2]);
    lo.z = __builtin_bswap32(t[1]); lo.w = __builtin_bswap32(t[0]);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = hi; q[1] = lo;
}

template <class M>
__device__ __forceinline__ Affine<M> load_affine(const Affine<M>* pts, uint32_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(pts + idx);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    Affine<M> r;
    r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w;
    r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
    r.y.v[0] = c.x; r.y.v[1] = c.y; r.y.v[2] = c.z; r.y.v[3] = c.w;
    r.y.v[4] = d.x; r.y.v[5] = d.y; r.y.v[6] = d.z; r.y.v[7] = d.w;
    return r;
}
template <class M>
__device__ __forceinline__ void store_fe(uint32_t* dst, const Fe<M>& f) {
    uint4* q = reinterpret_cast<uint4*>(dst);
    q[0] = make_uint4(f.v[0], f.v[1], f.v[2], f.v[3]);
    q[1] = make_uint4(f.v[4], f.v[5], f.v[6], f.v[7]);
}
template <class M>
__device__ __forceinline__ Fe<M> load_fe(const uint32_t* src) {
    const uint4* q = reinterpret_cast<const uint4*>(src);
    uint4 a = q[0], b = q[1];
    Fe<M> f;
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    return f;
}
template <class M>
__device__ __forceinline__ void store_xyzz(XYZZ<M>* dst, const XYZZ<M>& p) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    store_fe<M>(d, p.x); store_fe<M>(d + 8, p.y); store_fe<M>(d + 16, p.zz); store_fe<M>(d + 24, p.zzz);
}
template <class M>
__device__ __forceinline__ XYZZ<M> load_xyzz(const XYZZ<M>* src) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    XYZZ<M> p;
    p.x = load_fe<M>(s); p.y = load_fe<M>(s + 8); p.zz = load_fe<M>(s + 16); p.zzz = load_fe<M>(s + 24);
    return p;
}

// ------------------------------------------------------------------------------------------------
// G1Affine.Unmarshal semantics for the uncompressed form (main.go:130): X, Y <- SetBytes (reduced
// mod p); (0,0) stays (0,0) = infinity.  Output: Montgomery limbs, 64 B per point.
// F30: the residues are stored in the 2^270 Montgomery form of fe30.hip.h (still canonical 256-bit values) -- the MSM's
// bucket accumulation with C::F30_BUCKETS; every other consumer takes the 2^256 form.
template <class C, bool GLV, bool F30 = false>
__global__ void k_points_to_mont(const uint8_t* __restrict__ in, Affine<typename C::Fp>* __restrict__ out, uint32_t n) {
    using M = typename C::Fp;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<M> x, y;
    load_be256(x.v, in + (size_t)i * 64);
    load_be256(y.v, in + (size_t)i * 64 + 32);
    fe_reduce_plain<M>(x.v, 6);
    fe_reduce_plain<M>(y.v, 6);
    if constexpr (F30) {
        Fe<M> r2;
#pragma unroll
        for (int k = 0; k < 8; k++) r2.v[k] = M::R2_30[k];
        x = fe_mul<M>(x, r2);
        y = fe_mul<M>(y, r2);
    } else {
        x = fe_to_mont<M>(x);
        y = fe_to_mont<M>(y);
    }
    if (GLV) {
        // P at 2i, phi(P) = (beta * x, y) at 2i + 1; infinity (0, 0) stays (0, 0)
        Fe<M> beta;
#pragma unroll
        for (int k = 0; k < 8; k++) beta.v[k] = C::Glv::BETA[k];
        Fe<M> bx = fe_mul<M>(x, fe_to_mont<M>(beta));
        uint32_t* d = reinterpret_cast<uint32_t*>(out + 2 * (size_t)i);
        store_fe<M>(d, x);
        store_fe<M>(d + 8, y);
        store_fe<M>(d + 16, bx);
        store_fe<M>(d + 24, y);
    } else {
        uint32_t* d = reinterpret_cast<uint32_t*>(out + i);
        store_fe<M>(d, x);
        store_fe<M>(d + 8, y);
    }
}

// OR of all scalars, limb by limb (big-endian 32-byte scalars): tells the host how many windows are really needed.  The
// audit's coefficients are abs(int32) (bn254_scalar_set_int, utils.h:271-275; Server.hpp:617-621): 2 windows instead of 16.
static __global__ void __launch_bounds__(256)
k_scalar_or(const uint8_t* __restrict__ scalars, uint32_t n, uint32_t* __restrict__ out8) {
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t t[8];
        load_be256(t, scalars + (size_t)i * 32);
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] |= t[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t v = acc[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicOr(&out8[k], v);
    }
}

// ------------------------------------------------------------------------------------------------
// Two-pass LDS counting sort of the (window, bucket) keys -- no global atomics on the data path.
//
// Pass A  k_digits_partition: a block owns a TILE of 4096 scalars.  It reduces them mod the group order once
//         (fr.Element.SetBytes, main.go:127), keeps the limbs in registers, and for every window w emits the tile's
//         non-zero signed digits grouped by PARTITION (= LOW bits of the bucket id, so that a sparsely filled top window
//         still spreads over all partitions) into tile_items[w][tile][*] together with the offset table tile_off[w][tile][*].
//         An item is (bucket >> log2 P) | sign << lowbits | local_index << (lowbits + 1)   (local_index < 8192).
// Pass B  k_partition_sort: a block owns one (window, partition) = 2^lowbits buckets (1024 for c <= 18).  Its waves walk
//         that partition's run in every tile with COALESCED loads (a run is ~TILE / P items = 512 B at c = 16), twice:
//         first to count per bucket in LDS (-> counts, starts; the partition's slot range comes from ONE cursor atomic
//         per block), then to place the point indices with LDS cursors.
constexpr int TILE = 4096;
constexpr int TILE_THREADS = 1024;
constexpr int TILE_SPT = TILE / TILE_THREADS;  // scalars per thread
constexpr int MAX_PARTS = 128;                 // partitions per window
constexpr int SORT_LOWBITS = 10;               // buckets per partition = 2^10 unless that needs more than MAX_PARTS partitions
constexpr int SORT_MAX_LOW = 4096;             // LDS counters of k_partition_sort (lowbits <= 12)
constexpr int SORT_STAGE_CAP = 34816;          // indices a block can stage in LDS before writing them out coalesced (136 KiB)

// m = entries per window at most.  Wider partitions (up to 2^12 buckets) while a partition's expected entries still fit the
// LDS staging area: fewer, longer tile runs (c = 17 at 2^20 pairs: 32 partitions with 512-B runs instead of 64 with 256 B)
static inline int sort_lowbits(int c, size_t m, int W) {
    int lb = (c - 1) < SORT_LOWBITS ? (c - 1) : SORT_LOWBITS;
    // (only while the grid keeps at least 256 blocks: small inputs want the parallelism, not the longer runs)
    while (lb < c - 1 && lb < 12 && (m >> (c - 1 - lb - 1)) <= 32768 && ((size_t)W << (c - 1 - lb - 1)) >= 256) lb++;
    if (c - 1 - lb > 7) lb = c - 1 - 7;  // at most MAX_PARTS = 2^7 partitions
    return lb;
}

// exclusive scan of one value per thread over a 1024-thread block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* wsum /* 16 */, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t s = wsum[k];
        if ((uint32_t)k < wv) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// SUBS = 2 with GLV: every scalar k contributes the sub-scalars |k1| (point 2i) and |k2| (point 2i+1 = phi(P_i)), signs
// folded into the entry's sign bit; SUBS = 1: the scalar itself (point i).  An item's local index is j * SUBS + e.
template <class C, bool GLV>
__global__ void __launch_bounds__(TILE_THREADS)
k_digits_partition(const uint8_t* __restrict__ scalars, uint32_t n, int c, int W, int lowbits,
                   uint32_t* __restrict__ tile_items, uint16_t* __restrict__ tile_off, uint32_t* __restrict__ ctrl) {
    constexpr int SUBS = GLV ? 2 : 1;
    constexpr int LIMBS = GLV ? 4 : 8;
    constexpr int NSUB = TILE_SPT * SUBS;
    __shared__ uint32_t hist[MAX_PARTS];
    __shared__ uint32_t cursor[MAX_PARTS];
    __shared__ uint32_t stage[TILE * SUBS];
    const uint32_t tile = blockIdx.x, T = gridDim.x, tid = threadIdx.x;
    // the control words of the kernels behind this one (sort cursor, item counters: see CTRL_WORDS below) start at zero -- cleared
    // here instead of by a memset packet in front of the chain (a 9-us fill kernel on a lone caller's critical path)
    if (tile == 0 && blockIdx.y == 0 && tid < 4 + 128) ctrl[tid] = 0;
    const int P = 1 << (c - 1 - lowbits);
    const uint32_t B = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    uint32_t t[NSUB][LIMBS];
    bool valid[TILE_SPT];
    uint32_t sneg = 0;  // bit u: sub-scalar u is negative
#pragma unroll
    for (int j = 0; j < TILE_SPT; j++) {
        uint32_t i = tile * TILE + j * TILE_THREADS + tid;
        valid[j] = i < n;
        uint32_t k[8];
#pragma unroll
        for (int q = 0; q < 8; q++) k[q] = 0;
        if (valid[j]) {
            load_be256(k, scalars + (size_t)i * 32);
            for (int q = 0; q < C::MAX_Q; q++) {
                uint32_t s[8];
                uint32_t br = 0;
#pragma unroll
                for (int q2 = 0; q2 < 8; q2++) s[q2] = sbb32(k[q2], C::ORDER[q2], br);
                if (br) break;
#pragma unroll
                for (int q2 = 0; q2 < 8; q2++) k[q2] = s[q2];
            }
        }
        if (GLV) {
            uint32_t m1[4], m2[4];
            bool n1, n2;
            glv_split<typename C::Glv>(k, m1, n1, m2, n2);
#pragma unroll
            for (int q = 0; q < 4; q++) { t[j * SUBS][q] = m1[q]; t[j * SUBS + (SUBS - 1)][q] = m2[q]; }
            sneg |= (n1 ? 1u : 0u) << (j * SUBS);
            sneg |= (n2 ? 1u : 0u) << (j * SUBS + (SUBS - 1));
        } else {
#pragma unroll
            for (int q = 0; q < LIMBS; q++) t[j][q] = k[q];
        }
    }
    uint32_t carry = 0;  // bit u: carry into the next window of sub-scalar u

    // gridDim.y blocks share a tile: block y emits the windows w = y (mod gridDim.y); the signed-digit carries of the
    // windows in between are still walked (cheap), the LDS grouping only runs for the block's own windows -- a small input
    // is one tile, and 17 windows in sequence were 70 us of a 0.5 ms MSM
    const bool shared_tile = gridDim.y > 1;
    for (int w = 0; w < W; w++) {
        const bool mine = !shared_tile || (uint32_t)w % gridDim.y == blockIdx.y;
        if (mine) {
            if (tid < MAX_PARTS) hist[tid] = 0;
            __syncthreads();
        }
        uint32_t key[NSUB];
        const int lo = w * c;
        const int limb = lo >> 5, sh = lo & 31;
        // the two words the window straddles: `limb` is the same for the whole block, so this is a scalar branch to one of
        // LIMBS copies (a select chain over the limbs cost 2 * LIMBS instructions per sub-scalar and window)
        uint32_t wa[NSUB], wb[NSUB];
#pragma unroll
        for (int u = 0; u < NSUB; u++) { wa[u] = 0; wb[u] = 0; }
#define PORLA_DIGIT_WORDS(k)                                                                              \
        if constexpr ((k) < LIMBS) {                                                                          \
            if (limb == (k)) {                                                                                \
                _Pragma("unroll") for (int u = 0; u < NSUB; u++) {                                            \
                    wa[u] = t[u][(k)];                                                                        \
                    wb[u] = ((k) + 1 < LIMBS) ? t[u][((k) + 1 < LIMBS) ? (k) + 1 : (k)] : 0u;                 \
                }                                                                                             \
            }                                                                                                 \
        }
        PORLA_DIGIT_WORDS(0) PORLA_DIGIT_WORDS(1) PORLA_DIGIT_WORDS(2) PORLA_DIGIT_WORDS(3)
        PORLA_DIGIT_WORDS(4) PORLA_DIGIT_WORDS(5) PORLA_DIGIT_WORDS(6) PORLA_DIGIT_WORDS(7)
#undef PORLA_DIGIT_WORDS
#pragma unroll
        for (int u = 0; u < NSUB; u++) {
            uint32_t raw = 0;
            if (lo < 32 * LIMBS) {
                uint64_t v = ((uint64_t)wb[u] << 32) | wa[u];
                raw = (uint32_t)(v >> sh) & mask;
            }
            raw += (carry >> u) & 1u;
            uint32_t dneg = 0;
            if (raw > B) {  // negative digit raw - 2^c, magnitude 1 .. B-1
                carry |= 1u << u;
                key[u] = ((1u << c) - raw) - 1;
                dneg = 1;
            } else {
                carry &= ~(1u << u);
                key[u] = raw ? (raw - 1) : KEY_NONE;
            }
            if (!valid[u / SUBS]) key[u] = KEY_NONE;
            if (mine && key[u] != KEY_NONE) {
                key[u] |= (dneg ^ ((sneg >> u) & 1u)) << 31;
                atomicAdd(&hist[(key[u] & 0x7fffffffu) & (uint32_t)(P - 1)], 1u);
            }
        }
        if (!mine) continue;
        __syncthreads();
        // exclusive scan over P <= 128 partitions: two waves scan their 64 counters with shuffles, one barrier to pass the first
        // wave's total on (a Hillis-Steele scan in LDS took 14 barriers of a 1024-thread block per window)
        uint32_t pv = 0, pincl = 0;
        if (tid < MAX_PARTS) {
            pv = (tid < (uint32_t)P) ? hist[tid] : 0;
            pincl = pv;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(pincl, d, 64);
                if ((tid & 63u) >= (uint32_t)d) pincl += o;
            }
            if (tid == 63) cursor[0] = pincl;
        }
        __syncthreads();
        uint16_t* off = tile_off + ((size_t)w * T + tile) * (MAX_PARTS + 1);
        if (tid < MAX_PARTS) {
            const uint32_t excl = (tid >= 64 ? cursor[0] : 0u) + pincl - pv;
            off[tid] = (uint16_t)excl;
            hist[tid] = excl;  // becomes the running cursor
            if (tid == MAX_PARTS - 1) { off[MAX_PARTS] = (uint16_t)(excl + pv); cursor[1] = excl + pv; }
        }
        __syncthreads();
        const uint32_t total = cursor[1];
#pragma unroll
        for (int u = 0; u < NSUB; u++) {
            if (key[u] != KEY_NONE) {
                uint32_t bkt = key[u] & 0x7fffffffu;
                uint32_t pos = atomicAdd(&hist[bkt & (uint32_t)(P - 1)], 1u);
                uint32_t local = (uint32_t)((u / SUBS) * TILE_THREADS + tid) * SUBS + (u % SUBS);
                stage[pos] = (bkt >> (c - 1 - lowbits)) | ((key[u] >> 31) << lowbits) | (local << (lowbits + 1));
            }
        }
        __syncthreads();
        uint32_t* dst = tile_items + ((size_t)w * T + tile) * (TILE * SUBS);
        for (uint32_t i = tid; i < total; i += TILE_THREADS) dst[i] = stage[i];
        __syncthreads();
    }
}

// grid = W * P blocks of 1024 threads (16 waves; wave v takes tiles v, v+16, ...).  STAGE_CAP / MAX_LOW: the staging area and the
// bucket counters in LDS -- the full size (one block per compute unit) or half of each (two: msm_impl.hip.h)
template <int STAGE_CAP, int MAX_LOW>
__global__ void __launch_bounds__(1024)
k_partition_sort(const uint32_t* __restrict__ tile_items, const uint16_t* __restrict__ tile_off, uint32_t T, uint32_t tile_cap,
                 int c, int lowbits, uint32_t* __restrict__ counts, uint32_t* __restrict__ starts,
                 uint32_t* __restrict__ entries, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t cnt[MAX_LOW];
    __shared__ uint32_t stage[STAGE_CAP];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t base_sh, total_sh;
    const int P = 1 << (c - 1 - lowbits);
    const uint32_t nlow = 1u << lowbits;
    const uint32_t w = blockIdx.x / P, p = blockIdx.x % P;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t lowmask = nlow - 1;
    for (uint32_t i = tid; i < nlow; i += 1024) cnt[i] = 0;
    __syncthreads();
    // A wave walks the runs of its tiles t = wv, wv + 16, ...: offsets -> items -> LDS atomics is a chain of two global loads
    // per run, so the run bounds are fetched two tiles ahead and the first 128 items of a run one tile ahead (runs are 128
    // items at 2^20 pairs): the loads of the next tiles are in flight while this one is counted / placed.
    auto run_bounds = [&](uint32_t t, uint32_t& lo, uint32_t& hi) {
        lo = 0; hi = 0;
        if (t < T) {
            const uint16_t* off = tile_off + ((size_t)w * T + t) * (MAX_PARTS + 1);
            lo = off[p];
            hi = (p + 1 < (uint32_t)P) ? off[p + 1] : off[MAX_PARTS];
        }
    };
    auto run_items = [&](uint32_t t, uint32_t lo, uint32_t hi, uint32_t& a, uint32_t& b) {
        a = 0; b = 0;
        if (t < T) {
            const uint32_t* it = tile_items + ((size_t)w * T + t) * tile_cap;
            if (lo + lane < hi) a = it[lo + lane];
            if (lo + 64 + lane < hi) b = it[lo + 64 + lane];
        }
    };
    auto count_chunk = [&](bool active, uint32_t item) {
        const uint32_t low = active ? (item & lowmask) : 0xffffffffu;
        // skewed digits (a sparsely populated top window): lanes that hit the first lane's bucket share one atomic
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)low);
        const unsigned long long same = __ballot(low == first);
        if (__popcll(same) >= 8) {
            if (lane == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&cnt[first], (uint32_t)__popcll(same));
            if (active && low != first) atomicAdd(&cnt[low], 1u);
        } else if (active) {
            atomicAdd(&cnt[low], 1u);
        }
    };
    {
        uint32_t lo1, hi1, lo2, hi2, a1, b1;
        run_bounds(wv, lo1, hi1);
        run_bounds(wv + 16, lo2, hi2);
        run_items(wv, lo1, hi1, a1, b1);
        for (uint32_t t = wv; t < T; t += 16) {
            const uint32_t lo = lo1, hi = hi1, a = a1, b = b1;
            lo1 = lo2; hi1 = hi2;
            run_bounds(t + 32, lo2, hi2);
            run_items(t + 16, lo1, hi1, a1, b1);
            const uint32_t* it = tile_items + ((size_t)w * T + t) * tile_cap;
            for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
                const bool active = i0 + lane < hi;
                const uint32_t item = i0 == lo ? a : (i0 == lo + 64 ? b : (active ? it[i0 + lane] : 0u));
                count_chunk(active, item);
            }
        }
    }
    __syncthreads();
    // exclusive scan over the nlow counters: thread owns `per` consecutive ones
    const uint32_t per = nlow > 1024 ? nlow / 1024 : 1;
    const uint32_t first = tid * per;
    uint32_t mine[4] = {0, 0, 0, 0};
    uint32_t sum = 0;
    if (first < nlow) {
        for (uint32_t k = 0; k < per; k++) { mine[k] = cnt[first + k]; sum += mine[k]; }
    }
    uint32_t total;
    uint32_t excl = block_scan_1024(sum, wsum, &total);
    if (tid == 0) { base_sh = atomicAdd(cursor, total); total_sh = total; }
    __syncthreads();
    // the partition's indices are staged in LDS and written out as one coalesced range when they fit (scattered 4-byte
    // stores cost ~6.5x their size in HBM write traffic); larger partitions store directly
    const bool staged = total_sh <= (uint32_t)STAGE_CAP;
    if (first < nlow) {
        uint32_t run = base_sh + excl;
        for (uint32_t k = 0; k < per; k++) {
            size_t b = ((size_t)w << (c - 1)) + ((size_t)(first + k) << (c - 1 - lowbits)) + p;
            counts[b] = mine[k];
            starts[b] = run;
            cnt[first + k] = run;  // running cursor per bucket
            run += mine[k];
        }
    }
    __syncthreads();
    uint32_t lo1, hi1, lo2, hi2, a1, b1;
    run_bounds(wv, lo1, hi1);
    run_bounds(wv + 16, lo2, hi2);
    run_items(wv, lo1, hi1, a1, b1);
    for (uint32_t t = wv; t < T; t += 16) {
        const uint32_t lo = lo1, hi = hi1, a = a1, b = b1;
        lo1 = lo2; hi1 = hi2;
        run_bounds(t + 32, lo2, hi2);
        run_items(t + 16, lo1, hi1, a1, b1);
        const uint32_t* it = tile_items + ((size_t)w * T + t) * tile_cap;
        for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
            const bool active = i0 + lane < hi;
            const uint32_t item = i0 == lo ? a : (i0 == lo + 64 ? b : (active ? it[i0 + lane] : 0u));
            const uint32_t low = active ? (item & lowmask) : 0xffffffffu;
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)low);
            const unsigned long long same = __ballot(low == first);
            uint32_t pos = 0;
            if (__popcll(same) >= 8) {
                const uint32_t leader = (uint32_t)(__ffsll((long long)same) - 1);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&cnt[first], (uint32_t)__popcll(same));
                base = __shfl(base, (int)leader, 64);
                if (low == first) pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                else if (active) pos = atomicAdd(&cnt[low], 1u);
            } else if (active) {
                pos = atomicAdd(&cnt[low], 1u);
            }
            if (active) {
                const uint32_t val = (t * tile_cap + (item >> (lowbits + 1))) | (((item >> lowbits) & 1u) << 31);
                if (staged) stage[pos - base_sh] = val;
                else entries[pos] = val;
            }
        }
    }
    if (staged) {
        __syncthreads();
        for (uint32_t i = tid; i < total_sh; i += 1024) entries[base_sh + i] = stage[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Scheduling of the bucket accumulation.  A WORK ITEM is at most CHUNK consecutive entries of one bucket: a bucket with
// cnt entries yields cnt / CHUNK full items and one remainder item, so a skewed input (all scalars equal, or the few
// occupied buckets of a partially filled top window) becomes many items instead of one 2^20-long dependent chain.  Items
// are counting-sorted by size, largest first, so that the 64 lanes of a wave run (almost) the same trip count.
//   ctrl[0] entries cursor   ctrl[1] chunk-output cursor   ctrl[2] number of multi-item buckets   ctrl[3] number of items
//   ctrl[4 .. 4+CHUNK)  items per size row (row r <-> size CHUNK - r)
constexpr int CHUNK = 128;
constexpr int CTRL_WORDS = 4 + CHUNK;   // (k_digits_partition clears 4 + 128 words)
static_assert(CTRL_WORDS == 4 + 128, "k_digits_partition clears the control words");
constexpr uint32_t NO_CHUNK = 0xffffffffu;

// per-1024-bucket block: histogram of item sizes; layout row-major [row][block], row = CHUNK - size
static __global__ void __launch_bounds__(1024)
k_size_hist(const uint32_t* __restrict__ counts, uint32_t nb, uint32_t* __restrict__ blk_hist, uint32_t nblocks,
            uint32_t* __restrict__ ctrl) {
    __shared__ uint32_t hist[CHUNK];
    if (threadIdx.x < CHUNK) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) {
        uint32_t cnt = counts[i];
        uint32_t full = cnt / CHUNK, rem = cnt % CHUNK;
        if (full) atomicAdd(&hist[0], full);
        if (rem) atomicAdd(&hist[CHUNK - rem], 1u);
    }
    __syncthreads();
    if (threadIdx.x < CHUNK) {
        uint32_t v = hist[threadIdx.x];
        blk_hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = v;   // (row totals: k_size_scan sums its row, no global atomics)
    }
}

// grid = CHUNK blocks (one per size row): blk_off[row][blk] = items of larger sizes + items of this size in earlier blocks
static __global__ void __launch_bounds__(1024)
k_size_scan(const uint32_t* __restrict__ blk_hist, uint32_t* __restrict__ blk_off, uint32_t nblocks,
            uint32_t* __restrict__ ctrl) {
    __shared__ uint32_t wsum[16];
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    // offsets inside the row; the row's total goes to ctrl[4 + row] and k_size_order adds the rows before it
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024) {
        uint32_t x = (b0 + tid < nblocks) ? blk_hist[(size_t)row * nblocks + b0 + tid] : 0;
        uint32_t tot;
        uint32_t e = block_scan_1024(x, wsum, &tot);
        if (b0 + tid < nblocks) blk_off[(size_t)row * nblocks + b0 + tid] = run + e;
        run += tot;
        __syncthreads();
    }
    if (tid == 0) ctrl[4 + row] = run;
}

// order[pos] = (bucket, item index inside the bucket); empty buckets are written as infinity (zz = 0) here (buckets_raw ==
// nullptr: the bucket array already holds sums this launch adds to, and empty buckets keep theirs).
static __global__ void __launch_bounds__(1024)
k_size_order(const uint32_t* __restrict__ counts, uint32_t nb, const uint32_t* __restrict__ blk_off, uint32_t nblocks,
             uint2* __restrict__ order, uint32_t* __restrict__ chunk_base, uint32_t* __restrict__ heavy_list,
             uint32_t* __restrict__ ctrl, uint4* __restrict__ buckets_raw) {
    __shared__ uint32_t next[CHUNK];
    __shared__ uint32_t wsum[16];
    {
        // rows before this one (larger items): exclusive scan of the CHUNK row totals
        uint32_t total;
        const uint32_t v = (threadIdx.x < CHUNK) ? ctrl[4 + threadIdx.x] : 0;
        const uint32_t excl = block_scan_1024(v, wsum, &total);
        if (threadIdx.x < CHUNK) next[threadIdx.x] = excl + blk_off[(size_t)threadIdx.x * nblocks + blockIdx.x];
        if (blockIdx.x == 0 && threadIdx.x == 0) ctrl[3] = total;
    }
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    uint32_t cnt = counts[i];
    uint32_t full = cnt / CHUNK, rem = cnt % CHUNK;
    uint32_t items = full + (rem ? 1u : 0u);
    if (items == 0 && buckets_raw) {
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4* d = buckets_raw + (size_t)i * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = z;
    }
    if (full) {
        uint32_t pos = atomicAdd(&next[0], full);
        for (uint32_t j = 0; j < full; j++) order[pos + j] = make_uint2(i, j);
    }
    if (rem) {
        uint32_t pos = atomicAdd(&next[CHUNK - rem], 1u);
        order[pos] = make_uint2(i, full);
    }
    if (items > 1) {
        chunk_base[i] = atomicAdd(&ctrl[1], items);
        heavy_list[atomicAdd(&ctrl[2], 1u)] = i;
    } else {
        chunk_base[i] = NO_CHUNK;
    }
}

// ONE THREAD PER WORK ITEM, items taken in size order.
// The accumulation in the reduced-radix field form (ec30.hip.h): the points arrive in the 2^270 Montgomery form
// (k_points_to_mont<.., F30>), the item's sum leaves in the lazy memory form of ec30.hip.h, which the combine and tree kernels
// of this curve read; the last tree level converts to the 2^256 form for the host.
template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C::BUCKET_SUM_WAVES, 4)))
k_bucket_sum30(const Affine<typename C::Fp>* __restrict__ pts, const uint32_t* __restrict__ entries,
               const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
               const uint2* __restrict__ order, const uint32_t* __restrict__ chunk_base,
               const uint32_t* __restrict__ ctrl, XYZZ<typename C::Fp>* __restrict__ buckets,
               XYZZ<typename C::Fp>* __restrict__ chunk_out, uint32_t accumulate) {
    using M = typename C::Fp;
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= ctrl[3]) return;
    uint2 item = order[tid];
    const uint32_t b = item.x;
    const uint32_t first = item.y * CHUNK;
    uint32_t cnt = counts[b] - first;
    if (cnt > CHUNK) cnt = CHUNK;
    const uint32_t* e = entries + starts[b] + first;
    const uint32_t cb = chunk_base[b];
    XYZZ30<M> acc = xyzz30_infinity<M>();
    // accumulate: the bucket array holds the sums of earlier pair ranges of the same input (msm_host_multi) -- the bucket's only
    // item continues from that sum (multi-item buckets: k_bucket_combine adds it)
    if constexpr (C::F30_LAZY) {
        if (accumulate && cb == NO_CHUNK) acc = xyzz30_load_lazy<M>(buckets + b);
    }
    // the accumulator changes sign with every addition (ec30.hip.h:xyzz30_madd_flip: Y3 from ONE two-product reduction); the sign
    // it currently carries is folded into the digit's sign of the incoming point
    bool flip = false;
    uint32_t ent = e[0];
    uint32_t k = 0;
    if (acc.inf) {
        // the item's first entry is a copy, taken here so that the loop's straight-line form does not meet an accumulator at infinity
        // once per item
        const uint32_t cur = ent;
        if (cnt > 1) ent = e[1];
        Affine<M> a = load_affine<M>(pts, cur & 0x7fffffffu);
        if (!aff_is_inf<M>(a)) {
            a = aff_neg_if<M>(a, (cur >> 31) != 0);
            acc.x = f30_from_fe<M>(a.x); acc.y = f30_from_fe<M>(a.y);
            acc.zz = f30_const<M>(M::R1_30); acc.zzz = acc.zz;
            acc.inf = false;
        }
        k = 1;
        // the item's second entry meets an accumulator that is still an affine point: the short form (ec30.hip.h:xyzz30_mmadd_flip_fast).
        // An exceptional case leaves everything as it was and the loops below take the entry
        if (!acc.inf && cnt > 1) {
            const uint32_t cur2 = ent;
            uint32_t nxt = cur2;
            if (cnt > 2) nxt = e[2];
            Affine<M> a2 = load_affine<M>(pts, cur2 & 0x7fffffffu);
            const bool a2_inf = aff_is_inf<M>(a2);
            a2 = aff_neg_if<M>(a2, xyzz30_flip_neg<M>((cur2 >> 31) != 0, flip));
            if (xyzz30_mmadd_flip_fast<M>(acc, flip, f30_from_fe<M>(a2.x), f30_from_fe<M>(a2.y), a2_inf)) { ent = nxt; k = 2; }
        }
    }
    {
        // the fast loop: straight-line additions; a lane that meets an exceptional case leaves it with that entry still to do
        for (; k < cnt; k++) {
            const uint32_t cur = ent;
            uint32_t nxt = cur;
            if (k + 1 < cnt) nxt = e[k + 1];           // the next index is in flight while this point is accumulated
            Affine<M> a = load_affine<M>(pts, cur & 0x7fffffffu);
            const bool a_inf = aff_is_inf<M>(a);
            a = aff_neg_if<M>(a, xyzz30_flip_neg<M>((cur >> 31) != 0, flip));
            if (!xyzz30_madd_flip_fast<M>(acc, flip, f30_from_fe<M>(a.x), f30_from_fe<M>(a.y), a_inf)) break;
            ent = nxt;
        }
    }
    // the general form: every exceptional case, for the entries the fast loop left (none for uniformly random inputs)
    for (; k < cnt; k++) {
        const uint32_t cur = ent;
        if (k + 1 < cnt) ent = e[k + 1];
        Affine<M> a = load_affine<M>(pts, cur & 0x7fffffffu);
        if (aff_is_inf<M>(a)) continue;
        a = aff_neg_if<M>(a, xyzz30_flip_neg<M>((cur >> 31) != 0, flip));
        xyzz30_madd_flip<M>(acc, flip, f30_from_fe<M>(a.x), f30_from_fe<M>(a.y));
    }
    xyzz30_flip_finish<M>(acc, flip);
    XYZZ<M>* dst = cb == NO_CHUNK ? buckets + b : chunk_out + cb + item.y;
    if constexpr (C::F30_LAZY) xyzz30_store_lazy<M>(dst, acc);
    else store_xyzz<M>(dst, xyzz30_to_xyzz<M>(acc));
}


template <class M>
__device__ __forceinline__ XYZZ<M> xyzz_shfl_xor(const XYZZ<M>& p, int mask) {
    XYZZ<M> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.v[i] = __shfl_xor(p.x.v[i], mask, 64);
        r.y.v[i] = __shfl_xor(p.y.v[i], mask, 64);
        r.zz.v[i] = __shfl_xor(p.zz.v[i], mask, 64);
        r.zzz.v[i] = __shfl_xor(p.zzz.v[i], mask, 64);
    }
    return r;
}
// ------------------------------------------------------------------------------------------------
// Bucket reduction as a BIT-SLICED TREE: no scalar multiples, log2(B) dependent additions instead of the ~45 dependent
// group operations of the earlier running-sum form, and 2 additions per bucket in total.
//   sum_b (b+1) B_b = S + sum_k 2^k M_k,   S = sum of the window's buckets,  M_k = sum of the buckets whose index has bit k set.
// Level l = 0 .. c-2 halves the node count; node i of level l covers buckets [i 2^(l+1), (i+1) 2^(l+1)) and holds
//   S^l[i]   = S^(l-1)[2i] + S^(l-1)[2i+1]                       (S^(-1) = the buckets themselves)
//   M_k^l[i] = M_k^(l-1)[2i] + M_k^(l-1)[2i+1]     for k < l-1
//   M_(l-1)^l[i] = S^(l-2)[4i+1] + S^(l-2)[4i+3]                 (M_(l-1)^(l-1)[j] is S^(l-2)[2j+1]: an alias, never copied)
// so a level is (l+1) * nodes INDEPENDENT additions: one lane each.  The S levels are kept (their odd entries are read two
// levels later), the M slots ping-pong.  The last level writes fin[w][0] = S, fin[w][1+k] = M_k (k < c-1) and the host runs
// the Horner fold over single bits (h_fold_tree64): the W*c doublings it already did, W*c additions instead of W.
template <class M>
struct TreeLevelArgs {
    const XYZZ<M>* s_prev;   // S^(l-1): 2 n entries (the buckets at l = 0)
    const XYZZ<M>* s_prev2;  // S^(l-2): 4 n entries (the buckets at l = 1; unused at l = 0)
    const XYZZ<M>* m_prev;   // M^(l-1): slot k at m_prev + k * m_prev_stride, k < l-1
    XYZZ<M>* s_out;          // S^l: n entries
    XYZZ<M>* m_out;          // M^l: slot k at m_out + k * m_out_stride, k < l
    XYZZ<M>* fin;            // last level only: [W][nlev + 1]
    uint32_t n;              // nodes of this level, all windows together
    uint32_t m_prev_stride, m_out_stride;
    uint32_t l;
    uint32_t nlev;           // c - 1
    uint32_t last;           // 1: write fin instead of s_out / m_out (+ one copy task per node for the aliased M_l)
};

// How the reduction kernels read, add and write bucket sums: the lazy reduced-radix form of ec30.hip.h where the curve has
// it (C::F30_LAZY), ec.hip.h's XYZZ otherwise; `fin` (read by the host) is always XYZZ in the 2^256 form.
template <class C>
struct Node {
    using M = typename C::Fp;
    using T = typename std::conditional<C::F30_LAZY, XYZZ30<M>, XYZZ<M>>::type;
    static __device__ __forceinline__ T load(const XYZZ<M>* p) {
        if constexpr (C::F30_LAZY) return xyzz30_load_lazy<M>(p);
        else return load_xyzz<M>(p);
    }
    static __device__ __forceinline__ void add(T& a, const T& b) {
        if constexpr (C::F30_LAZY) xyzz30_add<M>(a, b);
        else xyzz_add_cold<M>(&a, &b);
    }
    static __device__ __forceinline__ void store(XYZZ<M>* p, const T& a) {
        if constexpr (C::F30_LAZY) xyzz30_store_lazy<M>(p, a);
        else store_xyzz<M>(p, a);
    }
    static __device__ __forceinline__ void store_final(XYZZ<M>* p, const T& a) {
        if constexpr (C::F30_LAZY) store_xyzz<M>(p, xyzz30_to_xyzz<M>(a));
        else store_xyzz<M>(p, a);
    }
    static __device__ __forceinline__ T inf() {
        if constexpr (C::F30_LAZY) { T r; r.x = r.y = r.zz = r.zzz = F30<M>{}; r.inf = true; return r; }
        else return xyzz_inf<M>();
    }
};

// one addition (or, at the last level, one copy) of level a.l; s = slot, i = node (global index: the S levels and fin),
// jp / jo = the node's index inside the m_prev / m_out slot arrays
template <class M>
struct TreeOp {
    const XYZZ<M>*pa, *pb;   // operands (copy: pa only)
    XYZZ<M>* out;
    bool final;              // out is fin (2^256 form for the host)
    bool copy;               // last level: the aliased top slot is copied into fin
};
template <class M>
__device__ __forceinline__ TreeOp<M> tree_op(const TreeLevelArgs<M>& a, uint32_t s, uint32_t i, uint32_t jp, uint32_t jo) {
    TreeOp<M> o;
    o.final = a.last != 0;
    o.copy = (s == a.l + 1);
    if (o.copy) {
        o.pa = a.s_prev + 2 * (size_t)i + 1; o.pb = o.pa;
        o.out = a.fin + (size_t)i * (a.nlev + 1) + 1 + a.l;
        return o;
    }
    if (s == a.l) { o.pa = a.s_prev + 2 * (size_t)i; o.pb = o.pa + 1; }
    else if (s + 1 == a.l) { o.pa = a.s_prev2 + 4 * (size_t)i + 1; o.pb = o.pa + 2; }
    else { o.pa = a.m_prev + (size_t)s * a.m_prev_stride + 2 * (size_t)jp; o.pb = o.pa + 1; }
    if (a.last) o.out = a.fin + (size_t)i * (a.nlev + 1) + (s == a.l ? 0 : 1 + s);
    else o.out = (s == a.l) ? a.s_out + i : a.m_out + (size_t)s * a.m_out_stride + jo;
    return o;
}
template <class C>
__device__ __forceinline__ void tree_task(const TreeLevelArgs<typename C::Fp>& a, uint32_t s, uint32_t i, uint32_t jp, uint32_t jo) {
    using M = typename C::Fp;
    using N = Node<C>;
    const TreeOp<M> o = tree_op<M>(a, s, i, jp, jo);
    typename N::T x = N::load(o.pa);
    if (!o.copy) {
        typename N::T y = N::load(o.pb);
        N::add(x, y);
    }
    if (o.final) N::store_final(o.out, x);
    else N::store(o.out, x);
}
// the same on the four lanes of a quad (ec30.hip.h:xyzz30_add_quad): ALL lanes of the quad call it with the same arguments
template <class C>
__device__ __forceinline__ void tree_task_quad(const TreeLevelArgs<typename C::Fp>& a, uint32_t s, uint32_t i, uint32_t jp, uint32_t jo,
                                               bool live, uint32_t lane) {
    using M = typename C::Fp;
    const TreeOp<M> o = tree_op<M>(a, s, i, jp, jo);
    if (o.copy) {                                     // uniform over the quad
        if (live && (lane & 3u) == 0u) Node<C>::store_final(o.out, Node<C>::load(o.pa));
        return;
    }
    xyzz30_add_quad<M>(o.pa, o.pb, o.out, o.final, live, lane);
}

// one level over all windows: grid covers (l + 1 + last) * n tasks, node index fastest (coalesced 256-B reads per lane pair)
template <class C>
__global__ void __launch_bounds__(256)
k_tree_level(TreeLevelArgs<typename C::Fp> a) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t slots = a.l + 1 + a.last;
    if (t >= slots * a.n) return;
    const uint32_t i = t % a.n;
    tree_task<C>(a, t / a.n, i, i, i);
}

// a level with few additions (latency bound): four lanes per addition
template <class C>
__global__ void __launch_bounds__(256)
k_tree_level_quad(TreeLevelArgs<typename C::Fp> a) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = (a.l + 1 + a.last) * a.n;
    uint32_t t = tid >> 2;
    const bool live = t < total;
    if (!live) t = total - 1;                         // padding quads compute a valid task and store nothing
    const uint32_t i = t % a.n;
    tree_task_quad<C>(a, t / a.n, i, i, i, live, threadIdx.x & 63u);
}

// the last levels l0 .. nlev-1 of ONE window per block (few nodes are left: launch gaps would dominate).  The levels
// exchange their nodes through global memory, a block-wide barrier between them.  Blocks run at their own pace, so the M
// slots of the tail are PRIVATE to the window (m_tail halves, per_window entries each per window) -- the all-window slot
// arrays of the level kernels interleave windows and change layout from level to level; only level l0 reads them.
// s_lev[l] = S^l (all windows: a node's address does not depend on the level another window is at).
template <class M>
struct TreeTailArgs {
    const XYZZ<M>* buckets;
    XYZZ<M>* s_lev[24];
    const XYZZ<M>* m_global;  // M^(l0-1) as written by k_tree_level (stride 2 * (nb >> (l0+1)))
    XYZZ<M>* m_tail[2];
    XYZZ<M>* fin;
    uint32_t nb;      // W * B
    uint32_t B;
    uint32_t per_window;
    uint32_t l0;
    uint32_t nlev;
};
template <class C, bool QUAD>
__global__ void __launch_bounds__(512)
k_tree_tail(TreeTailArgs<typename C::Fp> a) {
    using M = typename C::Fp;
    const uint32_t w = blockIdx.x;
    for (uint32_t l = a.l0; l < a.nlev; l++) {
        const uint32_t nw = a.B >> (l + 1);            // nodes of this window at level l
        TreeLevelArgs<M> lv;
        lv.s_prev = l ? a.s_lev[l - 1] : a.buckets;
        lv.s_prev2 = l >= 2 ? a.s_lev[l - 2] : a.buckets;
        lv.n = a.nb >> (l + 1);
        const bool first = (l == a.l0);
        lv.m_prev = first ? a.m_global : a.m_tail[(l + 1) & 1] + (size_t)w * a.per_window;
        lv.m_prev_stride = first ? 2 * lv.n : 2 * nw;
        lv.s_out = a.s_lev[l];
        lv.m_out = a.m_tail[l & 1] + (size_t)w * a.per_window;
        lv.m_out_stride = nw;
        lv.fin = a.fin;
        lv.l = l;
        lv.nlev = a.nlev;
        lv.last = (l + 1 == a.nlev) ? 1u : 0u;
        const uint32_t tasks = (l + 1 + lv.last) * nw;
        if constexpr (QUAD) {
            // four lanes per addition; every pass runs whole quads (padding quads redo the last task without storing)
            const uint32_t per_pass = blockDim.x >> 2;
            for (uint32_t t0 = 0; t0 < tasks; t0 += per_pass) {
                uint32_t t = t0 + (threadIdx.x >> 2);
                const bool live = t < tasks;
                if (!live) t = tasks - 1;
                const uint32_t j = t % nw, i = w * nw + j;
                tree_task_quad<C>(lv, t / nw, i, first ? i : j, j, live, threadIdx.x & 63u);
            }
        } else {
            for (uint32_t t = threadIdx.x; t < tasks; t += blockDim.x) {
                const uint32_t j = t % nw, i = w * nw + j;
                tree_task<C>(lv, t / nw, i, first ? i : j, j);
            }
        }
        __threadfence();
        __syncthreads();
    }
}

// wave per multi-item bucket: buckets[b] = sum of its item sums (ctrl[2] buckets listed in heavy_list)
template <class C>
__global__ void __launch_bounds__(64)
k_bucket_combine(const uint32_t* __restrict__ heavy_list, const uint32_t* __restrict__ chunk_base,
                 const uint32_t* __restrict__ counts, const uint32_t* __restrict__ ctrl,
                 const XYZZ<typename C::Fp>* __restrict__ chunk_out, XYZZ<typename C::Fp>* __restrict__ buckets,
                 uint32_t accumulate) {
    using M = typename C::Fp;
    using N = Node<C>;
    __shared__ XYZZ<M> stage[64];
    const uint32_t n_heavy = ctrl[2];
    for (uint32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        const uint32_t b = heavy_list[h];
        const uint32_t items = (counts[b] + CHUNK - 1) / CHUNK;
        const XYZZ<M>* src = chunk_out + chunk_base[b];
        typename N::T acc = N::inf();
        if (accumulate && threadIdx.x == 0) acc = N::load(buckets + b);      // the sums of earlier ranges (k_bucket_sum30)
        for (uint32_t k = threadIdx.x; k < items; k += 64) {
            typename N::T p = N::load(src + k);
            N::add(acc, p);
        }
        // fold the 64 lane sums through LDS in the memory form (one wave: no barrier needed beyond the LDS fence)
        for (uint32_t m = 32; m >= 1; m >>= 1) {
            N::store(&stage[threadIdx.x], acc);
            __syncthreads();
            if (threadIdx.x < m) {
                typename N::T o = N::load(&stage[threadIdx.x + m]);
                N::add(acc, o);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) N::store(buckets + b, acc);
    }
}

}  // namespace porla
