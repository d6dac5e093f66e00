// Launch sequence of the bucket MSM, templated on the curve; included by one translation unit per curve
// (msm_bn254.hip, msm_secp256k1.hip) so that the two instantiations compile in parallel.
#pragma once
#include "engine.hpp"
#include "host_fold64.hpp"
#include "msm_small.hip.h"
#include <cstdlib>
#include <chrono>
#include <thread>
#include <vector>
#include <cstdio>
#include "../../include/porla_gpu.h"

namespace porla {

static inline int ilog2(size_t n) { int l = 0; while (n >>= 1) l++; return l; }

// threads of the per-window tail block: the tail takes over at the first level whose additions per window fit one pass (four
// lanes per addition).  Same-box sweep of a blocking 2^20-pair MSM, profiles/r05_f_tree_tail_sweep.txt: 256 threads 1.58 ms,
// 512 threads 1.53-1.55, 1 024 threads 1.55-1.64 (one block per window then holds a compute unit for eight levels).
static inline uint32_t tree_tail_threads() { return 512u; }
// quad = four lanes per addition: a pass of the tail holds threads / 4 additions
static inline bool tree_quad() { return true; }
constexpr uint32_t TREE_QUAD_MAX_TASKS = 16384;         // levels up to this many additions run four lanes per addition ...
constexpr uint32_t TREE_QUAD_MAX_TASKS_LONE = 65536;    // ... for a caller that waits for this MSM alone (idle lanes to spend on latency; same sweep)
static inline uint32_t tree_tail_start(uint32_t B, uint32_t nlev, bool quad) {
    const uint32_t cap = quad ? tree_tail_threads() / 4 : tree_tail_threads();
    uint32_t l = 0;
    while (l + 1 < nlev && (l + 1) * (B >> (l + 1)) > cap) l++;
    return l;
}

// Window width for m sub-scalars of `bits` bits, from a time model in microseconds fitted to the sweeps in profiles/
// (2^7 .. 2^20 pairs):
//   digit extraction     ~6 us per window (the per-tile loop over windows)
//   bucket accumulation  throughput W*m / 13.5 G additions/s (reduced-radix form), but never faster than the longest dependent chain at ~5.6 us
//                        per entry for a lone wave: the average bucket plus 4 sigma, or -- the usual culprit -- a top window
//                        that holds only `top` bits and crowds all m entries into 2^(top-1) buckets (work items cap a chain
//                        at CHUNK entries)
//   bucket reduction     the tree's levels: throughput TREE_US_PER_ADD per addition, never below one launch + one
//                        dependent addition (TREE_LEVEL_US; TREE_QUAD_LEVEL_US with four lanes per addition);
//                        TREE_TAIL_LEVEL_US per level of the per-window tail kernel
constexpr double TREE_US_PER_ADD = 1.5e-4;     // 262 144 additions of level 0 at c = 16 in 39 us (reduced-radix form)
constexpr double TREE_LEVEL_US = 12.5;
constexpr double TREE_QUAD_LEVEL_US = 7.0;     // a level of <= TREE_QUAD_MAX_TASKS additions, four lanes per addition
constexpr double TREE_TAIL_LEVEL_US = 6.5;
static inline int choose_window(size_t m, int bits) {
    if (g_window_override >= 2 && g_window_override <= 20) return g_window_override;
    int best = 2;
    double best_cost = 1e300;
    for (int c = 2; c <= 20; c++) {
        const int W = (bits + 1 + c - 1) / c;
        const int top = bits + 1 - c * (W - 1);                       // bits left for the top window, 1 .. c
        const double B = (double)((size_t)1 << (c - 1));
        const double load = (double)m / B;                             // average entries per bucket
        const double top_load = W > 1 ? (double)m / (double)((size_t)1 << (top > 1 ? top - 1 : 0)) : 0.0;
        // ---- bucket accumulation: work items in waves of 64, 4096 waves resident at once (4 per SIMD).  With at least one
        // resident round the kernel is throughput bound: (entries - items) additions (an item's first entry is a copy) at 14 G/s,
        // stretched by the drain of the last round -- the fewer rounds, the larger its share (fitted at 2^20 pairs: c = 15 .. 18,
        // 1.06 .. 7.5 rounds, 1.53 / 1.30 / 1.08 / 0.95 ms).  Below one round the chains run closer to the lone-wave rate.
        const double items = (double)W * B * (load > 20.0 ? 1.0 : 1.0 - __builtin_exp(-load));
        const double waves = items / 64.0;
        const double rounds = waves / 4096.0;
        const double adds = (double)W * (double)m > items ? (double)W * (double)m - items : 0.0;
        double t_sum = rounds >= 1.0 ? adds / 14.0e3 * (1.0 + 0.22 / rounds)
                                     : (waves > 0.5 ? load * (4.0 + 15.4 * rounds) : 0.0);
        double chain = load + 4.0 * __builtin_sqrt(load) + 1.0;      // the longest dependent chain (lone wave: ~5.6 us / entry)
        const double top_chain = top_load < (double)CHUNK ? top_load : (double)CHUNK;
        if (top_chain > chain) chain = top_chain;
        if (chain > (double)CHUNK) chain = (double)CHUNK;
        if (t_sum < chain * 5.6) t_sum = chain * 5.6;
        // ---- heavy buckets: split into work items, one wave folds each bucket's item sums
        double t_combine = 0.0;
        if (top_load > (double)CHUNK) t_combine = 50.0 + top_load / CHUNK / 64.0 * 7.0;
        if (load > 0.6 * CHUNK) t_combine += 1e6;                      // every bucket would need the combine pass
        // ---- bucket reduction (bit-sliced tree, msm.hip.h): level l holds W (l+1) B / 2^(l+1) independent additions; a level
        // launch costs at least one addition's latency, the per-window tail levels run back to back in one kernel; the host
        // then folds W c single-bit terms
        double t_reduce = 0.3 * W * c;
        {
            const uint32_t Bu = 1u << (c - 1), nlev = (uint32_t)(c - 1);
            const uint32_t l0 = tree_tail_start(Bu, nlev, true);
            for (uint32_t l = 0; l < nlev; l++) {
                const double tasks = (double)W * (l + 1) * (double)(Bu >> (l + 1));
                if (l >= l0) t_reduce += TREE_TAIL_LEVEL_US;
                else if (tasks <= (double)TREE_QUAD_MAX_TASKS) t_reduce += TREE_QUAD_LEVEL_US;
                else { const double t = tasks * TREE_US_PER_ADD; t_reduce += t > TREE_LEVEL_US ? t : TREE_LEVEL_US; }
            }
        }
        // ---- LDS counting sort: ~6.6 ps per entry while a partition's buckets fit the staging area (c <= 16), slower beyond
        const double t_sort = (double)W * (double)m * 6.6e-6 * (c <= 16 ? 1.0 : (c == 17 ? 1.1 : 4.5));
        const double cost = 6.0 * W + t_sum + t_sort + t_combine + t_reduce;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

// the single-launch path of msm_small.hip.h (n <= SMALL_MAX_N): one kernel, results + shape written to pinned memory
template <class C>
static int msm_small_launch(Workspace* ws, const uint8_t* d_scalars, const uint8_t* d_points_be, size_t n, hipStream_t stream) {
    using M = typename C::Fp;
    int rc;
    if (ws->h_windows_cap < 64 * 1024) {
        if (ws->h_windows) PORLA_HIP(hipHostFree(ws->h_windows));
        ws->h_windows_cap = 64 * 1024;
        PORLA_HIP(hipHostMalloc(&ws->h_windows, ws->h_windows_cap, hipHostMallocMapped | hipHostMallocCoherent));
    }
    const size_t part_bytes = (size_t)SMALL_BLOCKS * SMALL_MAX_C * sizeof(XYZZ<M>);
    if (ws->small_part.cap < part_bytes + 2048) {    // 2 x 256 arrival counters (the pair form uses both halves)
        if ((rc = ws->small_part.ensure(part_bytes + 2048))) return rc;
        PORLA_HIP(hipMemsetAsync(ws->small_part.p, 0, part_bytes + 2048, stream));    // the per-window arrival counters start at zero
    }
    void* h_dev = nullptr;
    PORLA_HIP(hipHostGetDevicePointer(&h_dev, ws->h_windows, 0));
    uint32_t* counters = (uint32_t*)((uint8_t*)ws->small_part.p + part_bytes);
    ws->small_seq++;
    if (ws->small_seq == 0) ws->small_seq = 1;
    ((volatile uint32_t*)ws->h_windows)[0] = 0;
    {
        ProfScope ps("small_msm", stream, true);
        // up to 4 096 pairs 192 blocks are as fast as 256 (3 200 pairs: 0.1045 / 0.1046 ms with abs(int32) coefficients, 0.163 / 0.171
        // with 256-bit scalars; 16 384 pairs: 0.284 against 0.262) and leave a quarter of the chip to whatever short kernels run
        // beside this one
        const unsigned one_blocks = n <= 4096 ? 192u : (unsigned)SMALL_BLOCKS;
        hipLaunchKernelGGL((k_small_msm<C>), dim3(one_blocks), dim3(SMALL_THREADS), 0, stream, d_scalars, d_points_be, (uint32_t)n,
                           g_small_c | (g_use_glv == 0 ? 0x100 : 0), (XYZZ<M>*)ws->small_part.p, counters, (uint32_t*)h_dev,
                           (XYZZ<M>*)((uint8_t*)h_dev + SMALL_HDR_WORDS * 4), ws->small_seq, (const uint8_t*)nullptr, 0u);
    }
    PORLA_HIP(hipGetLastError());
    if (!ws->done) PORLA_HIP(hipEventCreateWithFlags(&ws->done, hipEventDisableTiming));
    PORLA_HIP(hipEventRecord(ws->done, stream));
    ws->pend_W = -1;            // shape known to the device only: read from the pinned header by msm_finish
    ws->pend_c = 0;
    return PORLA_OK;
}

// The audit's pair of MSMs -- ONE scalar array over TWO point arrays (bn254_multi_exp(combined_MAC, ptc, sc, n) and
// bn254_multi_exp(combined_align, pta, sc, n), porla/Server/Server.hpp:900-901; the IPA twins :842-848) -- as one launch of the
// single-launch kernel: half of the chip's blocks per point set.  Both are chains of ~15 dependent additions, so side by side
// they take little longer than one alone.  Results: two regions of the slot's pinned area, PAIR_STRIDE bytes apart.
constexpr uint32_t SMALL_PAIR_STRIDE = 64 * 1024;
template <class C>
static int msm_small_pair_launch(Workspace* ws, const uint8_t* d_scalars, const uint8_t* d_points_a, const uint8_t* d_points_b, size_t n,
                                 hipStream_t stream, int bits_hint = 0) {
    using M = typename C::Fp;
    int rc;
    if (ws->h_windows_cap < 2 * SMALL_PAIR_STRIDE) {
        if (ws->h_windows) PORLA_HIP(hipHostFree(ws->h_windows));
        ws->h_windows_cap = 2 * SMALL_PAIR_STRIDE;
        PORLA_HIP(hipHostMalloc(&ws->h_windows, ws->h_windows_cap, hipHostMallocMapped | hipHostMallocCoherent));
    }
    const size_t part_bytes = (size_t)SMALL_BLOCKS * SMALL_MAX_C * sizeof(XYZZ<M>);
    if (ws->small_part.cap < part_bytes + 2048) {
        if ((rc = ws->small_part.ensure(part_bytes + 2048))) return rc;
        PORLA_HIP(hipMemsetAsync(ws->small_part.p, 0, part_bytes + 2048, stream));    // arrival counters of both sets start at zero
    }
    void* h_dev = nullptr;
    PORLA_HIP(hipHostGetDevicePointer(&h_dev, ws->h_windows, 0));
    uint32_t* counters = (uint32_t*)((uint8_t*)ws->small_part.p + part_bytes);
    ws->small_seq++;
    if (ws->small_seq == 0) ws->small_seq = 1;
    ((volatile uint32_t*)ws->h_windows)[0] = 0;
    ((volatile uint32_t*)((uint8_t*)ws->h_windows + SMALL_PAIR_STRIDE))[0] = 0;
    {
        ProfScope ps("small_msm", stream, true);
        // blocks per point set: at the audit's sizes 96 (192 of the chip's 256 compute units) -- the sets are as fast as with 128 each
        // (0.120 against 0.124 ms at 3 200 pairs) and the audit's other chain, which runs beside this kernel (row combine, the
        // three commitments: short launches that need a compute unit NOW), no longer queues behind 256 long-lived blocks:
        // 0.183 -> 0.151 ms per audit
        const unsigned pair_blocks = n <= 8192 ? 96u : (unsigned)SMALL_BLOCKS / 2;
        hipLaunchKernelGGL((k_small_msm<C>), dim3(pair_blocks, 2), dim3(SMALL_THREADS), 0, stream, d_scalars, d_points_a, (uint32_t)n,
                           g_small_c | (g_use_glv == 0 ? 0x100 : 0) | (bits_hint << 16), (XYZZ<M>*)ws->small_part.p, counters,
                           (uint32_t*)h_dev, (XYZZ<M>*)((uint8_t*)h_dev + SMALL_HDR_WORDS * 4), ws->small_seq, d_points_b, SMALL_PAIR_STRIDE);
    }
    PORLA_HIP(hipGetLastError());
    if (!ws->done) PORLA_HIP(hipEventCreateWithFlags(&ws->done, hipEventDisableTiming));
    PORLA_HIP(hipEventRecord(ws->done, stream));
    return PORLA_OK;
}
// waits for set `which` (0, 1) of a pair launch and folds its window sums on the host
template <class C>
static int msm_small_pair_finish(Workspace* ws, int which, XYZZ<typename C::Fp>* total) {
    using M = typename C::Fp;
    const uint8_t* region = (const uint8_t*)ws->h_windows + (size_t)which * SMALL_PAIR_STRIDE;
    const volatile uint32_t* hdr = (const volatile uint32_t*)region;
    const auto t_spin = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint32_t it = 0;; it++) {
        if (__atomic_load_n((const uint32_t*)&hdr[0], __ATOMIC_ACQUIRE) == ws->small_seq) { seen = true; break; }
        __builtin_ia32_pause();
        if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(2)) break;
    }
    if (!seen) PORLA_HIP(hipEventSynchronize(ws->done));
    const int W = (int)hdr[1], c = (int)hdr[2];
    if (hdr[0] != ws->small_seq || W < 1 || c < 1 || c > SMALL_MAX_C || (size_t)W * c * sizeof(XYZZ<M>) + SMALL_HDR_WORDS * 4 > SMALL_PAIR_STRIDE) {
        set_last_error("porla: the single-launch MSM pair left no valid result header");
        return PORLA_ERR_HIP;
    }
    g_last_shape[0] = c; g_last_shape[1] = W; g_last_shape[2] = (int)hdr[3];
    *total = h_fold_tree64<M>((const XYZZ<M>*)(region + SMALL_HDR_WORDS * 4), W, c);
    return PORLA_OK;
}

// The shape several ranges of one input share when their bucket sums are merged before ONE reduction (msm_host_multi): the
// split flag and the scalar length are those of the whole input, the window width is given.
struct MsmShape {
    bool glv;
    int bits, c, W;
};
// Split the scalars with the endomorphism?  The caller's choice (porla_gpu_set_msm_glv), else the curve's default -- and for a
// curve whose default is "no" still up to GLV_BELOW pairs: there the launch chain and the host fold weigh more than the
// accumulation, and the split halves the windows and the fold (BN254: 0.39 / 0.43 / 0.48 / 0.57 / 0.58 ms against 0.45 / 0.48 /
// 0.52 / 0.62 / 0.62 ms at 2^13 .. 2^17 pairs, 0.78 against 0.76 ms at 2^18: profiles/r02_l_glv_mid_sizes.jsonl)
template <class C>
static inline bool msm_use_glv(size_t n) {
    if (g_use_glv >= 0) return g_use_glv != 0;
    return C::GLV_DEFAULT || n <= C::GLV_BELOW;
}
template <class C>
static inline MsmShape msm_full_shape(size_t n_range) {
    MsmShape sh;
    sh.glv = msm_use_glv<C>(n_range);
    sh.bits = sh.glv ? C::Glv::BITS : C::SCALAR_BITS;
    sh.c = choose_window(sh.glv ? 2 * n_range : n_range, sh.bits);
    sh.W = (sh.bits + 1 + sh.c - 1) / sh.c;
    return sh;
}
template <class C>
static int msm_tree_launch(Workspace* ws, const XYZZ<typename C::Fp>* buckets, int c, int W, bool glv, hipStream_t stream);

// forced != nullptr: bucket sums only, in the given shape, into bucket_out; `accumulate`: bucket_out already holds the sums
// of earlier ranges and this range adds to them, after `after` (the previous range's last write to bucket_out) has happened.
// The caller runs msm_tree_launch once.  Otherwise the complete MSM up to the per-window sums in pinned memory.
template <class C>
static int msm_launch(Workspace* ws, const uint8_t* d_scalars, const uint8_t* d_points_be, size_t n, hipStream_t stream,
                      const MsmShape* forced = nullptr, XYZZ<typename C::Fp>* bucket_out = nullptr, bool accumulate = false,
                      hipEvent_t after = nullptr) {
    using M = typename C::Fp;
    ws->pend_W = 0;
    if (n == 0) return PORLA_OK;
    if (n >= (1ull << 30)) { set_last_error("porla: MSM length must be < 2^30 per call (range-split larger inputs)"); return PORLA_ERR_ARG; }
    if (!forced && n <= SMALL_MAX_N && g_small_mode != 0 && g_window_override == 0) return msm_small_launch<C>(ws, d_scalars, d_points_be, n, stream);
    // Above 2^22 pairs the window model would ask for 18 bits and more, where a partition of the counting sort no longer fits its
    // LDS staging area (scattered 4-byte stores: 5.5 of 26 ms at 2^24).  Instead the input goes through in ranges of <= 2^22
    // pairs that accumulate into ONE bucket array (as the ranges of msm_host_multi do) with one reduction at the end; the scratch
    // is that of a 2^22-pair MSM.
    if (!forced && n > MSM_RANGE_MAX && C::F30_LAZY && g_window_override == 0) {
        const size_t R = (n + MSM_RANGE_MAX - 1) / MSM_RANGE_MAX;
        const MsmShape shape = msm_full_shape<C>((n + R - 1) / R);
        int rc2;
        if ((rc2 = ws->multi_acc.ensure(((size_t)shape.W << (shape.c - 1)) * sizeof(XYZZ<M>)))) return rc2;
        XYZZ<M>* acc = (XYZZ<M>*)ws->multi_acc.p;
        for (size_t r = 0; r < R; r++) {
            const size_t lo = (size_t)((unsigned __int128)r * n / R), hi = (size_t)((unsigned __int128)(r + 1) * n / R);
            if ((rc2 = msm_launch<C>(ws, d_scalars + 32 * lo, d_points_be + 64 * lo, hi - lo, stream, &shape, acc, r > 0, nullptr))) return rc2;
        }
        return msm_tree_launch<C>(ws, acc, shape.c, shape.W, shape.glv, stream);
    }
    bool glv = forced ? forced->glv : msm_use_glv<C>(n);
    int bits = glv ? C::Glv::BITS : C::SCALAR_BITS;
    // Small and medium inputs: look at the scalars first.  If none of them exceeds b bits (and b is below the group order's
    // length, so SetBytes does not reduce anything) only ceil((b + 1) / c) windows exist -- the audit's abs(int32)
    // coefficients need 2 windows of 16 bits, not 16.  One extra launch and an 8-word read-back (~15 us).
    if (!forced && n <= MSM_SCAN_MAX) {
        if (ws->h_windows_cap < 64 * 1024) {
            if (ws->h_windows) PORLA_HIP(hipHostFree(ws->h_windows));
            ws->h_windows_cap = 64 * 1024;
            PORLA_HIP(hipHostMalloc(&ws->h_windows, ws->h_windows_cap, hipHostMallocMapped | hipHostMallocCoherent));
        }
        int rc0;
        if ((rc0 = ws->cursor.ensure((CTRL_WORDS + 8) * 4))) return rc0;
        uint32_t* d_or = (uint32_t*)ws->cursor.p + CTRL_WORDS;
        PORLA_HIP(hipMemsetAsync(d_or, 0, 32, stream));
        unsigned blocks = (unsigned)((n + 255) / 256);
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(k_scalar_or, dim3(blocks), dim3(256), 0, stream, d_scalars, (uint32_t)n, d_or);
        PORLA_HIP(hipMemcpyAsync(ws->h_windows, d_or, 32, hipMemcpyDeviceToHost, stream));
        PORLA_HIP(hipStreamSynchronize(stream));
        const uint32_t* h_or = (const uint32_t*)ws->h_windows;
        int top = 7;
        while (top > 0 && h_or[top] == 0) top--;
        int used = h_or[top] ? 32 * top + (32 - __builtin_clz(h_or[top])) : 1;
        if (glv && g_use_glv < 0 && used <= C::Glv::BITS) { glv = false; bits = C::SCALAR_BITS; }   // short scalars: nothing to split
        if (used < 250 && used < bits) bits = used;       // < 2^250 < both group orders: no reduction happens
    }
    const size_t n_sub = glv ? 2 * n : n;                 // sub-scalars = entries per window at most
    const uint32_t tile_cap = glv ? 2 * TILE : TILE;
    const int c = forced ? forced->c : choose_window(n_sub, bits);
    const int W = (bits + 1 + c - 1) / c;
    const uint32_t B = 1u << (c - 1);
    const size_t nb = (size_t)W * B;
    const uint32_t n32 = (uint32_t)n;

    int rc;
    if ((rc = ws->pts.ensure(n_sub * sizeof(Affine<M>)))) return rc;
    const uint32_t T_tiles = (uint32_t)((n + TILE - 1) / TILE);
    const uint32_t nblk = (uint32_t)((nb + 1023) / 1024);
    const size_t max_entries = (size_t)W * n_sub;
    // the sort's cursor, starts[] and the entry offsets are 32-bit
    if (max_entries >= (1ull << 32)) { set_last_error("porla: MSM too large for one launch (windows x sub-scalars >= 2^32): range-split it (porla_*_msm_host_multi)"); return PORLA_ERR_ARG; }
    const size_t max_items = nb + max_entries / CHUNK;          // every bucket: <= cnt/CHUNK full items + 1 remainder
    const size_t max_chunk_out = 2 * (max_entries / CHUNK) + 2; // multi-item buckets only: ceil(cnt/CHUNK) <= 2 cnt/CHUNK
    if ((rc = ws->keys.ensure((size_t)W * T_tiles * tile_cap * 4))) return rc;   // tile_items
    if ((rc = ws->tile_off.ensure((size_t)W * T_tiles * (MAX_PARTS + 1) * 2))) return rc;
    if ((rc = ws->entries.ensure(max_entries * 4))) return rc;
    if ((rc = ws->counts.ensure(nb * 4))) return rc;
    if ((rc = ws->starts.ensure(nb * 4))) return rc;
    if ((rc = ws->fill.ensure(nb * 4))) return rc;                            // chunk_base
    if ((rc = ws->cursor.ensure((CTRL_WORDS + 8) * 4))) return rc;            // ctrl (+ the scalar OR words)
    if ((rc = ws->order.ensure(max_items * sizeof(uint2)))) return rc;
    if ((rc = ws->heavy.ensure((max_entries / CHUNK + 2) * 4))) return rc;
    if ((rc = ws->chunk_out.ensure(max_chunk_out * sizeof(XYZZ<M>)))) return rc;
    if ((rc = ws->blk_hist.ensure((size_t)CHUNK * nblk * 4))) return rc;
    if ((rc = ws->blk_off.ensure((size_t)CHUNK * nblk * 4))) return rc;
    XYZZ<M>* bk = bucket_out;
    if (!bk) {
        if ((rc = ws->buckets.ensure(nb * sizeof(XYZZ<M>)))) return rc;
        bk = (XYZZ<M>*)ws->buckets.p;
    }
    uint32_t* ctrl = (uint32_t*)ws->cursor.p;     // cleared by k_digits_partition
    const Affine<M>* pts = (const Affine<M>*)ws->pts.p;
    // The conversion of the points is independent of the digit / sort chain and only the accumulation reads its output: for a
    // caller that waits for this one MSM it runs on the workspace's second stream beside that chain (with another MSM in flight
    // the other MSM already fills the gaps, as for the tree split below)
    const bool front_split = ws->lone && !forced && n >= ((size_t)1 << 18);
    hipStream_t pst = stream;
    if (front_split) {
        if (!ws->aux_stream) PORLA_HIP(hipStreamCreateWithFlags(&ws->aux_stream, hipStreamNonBlocking));
        if (!ws->front_fork_ev) PORLA_HIP(hipEventCreateWithFlags(&ws->front_fork_ev, hipEventDisableTiming));
        if (!ws->front_join_ev) PORLA_HIP(hipEventCreateWithFlags(&ws->front_join_ev, hipEventDisableTiming));
        PORLA_HIP(hipEventRecord(ws->front_fork_ev, stream));
        PORLA_HIP(hipStreamWaitEvent(ws->aux_stream, ws->front_fork_ev, 0));
        pst = ws->aux_stream;
    }
    {
        ProfScope ps("points_to_mont", pst);
        if (glv)
            hipLaunchKernelGGL((k_points_to_mont<C, true, C::F30_BUCKETS>), dim3((n32 + 255) / 256), dim3(256), 0, pst, d_points_be,
                               (Affine<M>*)ws->pts.p, n32);
        else
            hipLaunchKernelGGL((k_points_to_mont<C, false, C::F30_BUCKETS>), dim3((n32 + 255) / 256), dim3(256), 0, pst, d_points_be,
                               (Affine<M>*)ws->pts.p, n32);
    }
    if (front_split) PORLA_HIP(hipEventRecord(ws->front_join_ev, ws->aux_stream));
    int lowbits = sort_lowbits(c, n_sub, W);
    // Half-size partitions -- twice the blocks, half the staging area (78 KiB of LDS): TWO sort blocks per compute unit, whose
    // load -> LDS-atomic chains then overlap -- where a partition's expected entries fit the smaller staging area and a tile's run
    // of a partition still holds 64 items (one coalesced load per wave).  Same box, profiles/r05_ag_sort_half_ab.txt: blocking calls of
    // 2^18 / 2^19 / 2^20 pairs -2 % / -2.3 % / -1.2 %, three in flight -1 % / -2.3 % / 0; 2^21 pairs (32-item runs) would lose 1.5 %.
    bool sort_half = false;
    {
        const int lb2 = lowbits - 1, parts_log = c - 1 - lb2;
        if (lb2 >= 0 && lb2 <= 11 && parts_log <= 7 && (n_sub >> parts_log) <= 16384 && (tile_cap >> parts_log) >= 64) {
            lowbits = lb2;
            sort_half = true;
        }
    }
    const int P = 1 << (c - 1 - lowbits);
    {
        ProfScope ps("digits_partition", stream);
        // blocks per tile: up to 2^19 pairs the windows of a tile are split over enough blocks to give every CU one (a block
        // walks its windows one after the other: 74 -> 50 us at 2^18 with four blocks per tile); larger inputs keep one block
        // per tile -- every extra block re-reads the tile's scalars: 0.07 -> 0.18 ms at 2^20 with four
        // (profiles/r02_g_digits_blocks_per_tile.jsonl)
        unsigned wg = T_tiles < 256 ? 256u / T_tiles : 1u;
        if (wg > 16) wg = 16;
        if (wg > (unsigned)W) wg = (unsigned)W;
        if (glv)
            hipLaunchKernelGGL((k_digits_partition<C, true>), dim3(T_tiles, wg), dim3(TILE_THREADS), 0, stream, d_scalars, n32, c, W,
                               lowbits, (uint32_t*)ws->keys.p, (uint16_t*)ws->tile_off.p, ctrl);
        else
            hipLaunchKernelGGL((k_digits_partition<C, false>), dim3(T_tiles, wg), dim3(TILE_THREADS), 0, stream, d_scalars, n32, c, W,
                               lowbits, (uint32_t*)ws->keys.p, (uint16_t*)ws->tile_off.p, ctrl);
    }
    {
        ProfScope ps("partition_sort", stream);
        if (sort_half)
            hipLaunchKernelGGL((k_partition_sort<SORT_STAGE_CAP / 2, SORT_MAX_LOW / 2>), dim3((unsigned)(W * P)), dim3(1024), 0, stream,
                               (const uint32_t*)ws->keys.p, (const uint16_t*)ws->tile_off.p, T_tiles, tile_cap, c, lowbits,
                               (uint32_t*)ws->counts.p, (uint32_t*)ws->starts.p, (uint32_t*)ws->entries.p, ctrl);
        else
            hipLaunchKernelGGL((k_partition_sort<SORT_STAGE_CAP, SORT_MAX_LOW>), dim3((unsigned)(W * P)), dim3(1024), 0, stream,
                               (const uint32_t*)ws->keys.p, (const uint16_t*)ws->tile_off.p, T_tiles, tile_cap, c, lowbits,
                               (uint32_t*)ws->counts.p, (uint32_t*)ws->starts.p, (uint32_t*)ws->entries.p, ctrl);
    }
    {
        ProfScope ps("size_order", stream);
        hipLaunchKernelGGL(k_size_hist, dim3(nblk), dim3(1024), 0, stream, (const uint32_t*)ws->counts.p, (uint32_t)nb,
                           (uint32_t*)ws->blk_hist.p, nblk, ctrl);
        hipLaunchKernelGGL(k_size_scan, dim3(CHUNK), dim3(1024), 0, stream, (const uint32_t*)ws->blk_hist.p,
                           (uint32_t*)ws->blk_off.p, nblk, ctrl);
        hipLaunchKernelGGL(k_size_order, dim3(nblk), dim3(1024), 0, stream, (const uint32_t*)ws->counts.p, (uint32_t)nb,
                           (const uint32_t*)ws->blk_off.p, nblk, (uint2*)ws->order.p, (uint32_t*)ws->fill.p,
                           (uint32_t*)ws->heavy.p, ctrl, accumulate ? (uint4*)nullptr : (uint4*)bk);
    }
    if (accumulate) {
        if (!C::F30_LAZY) { set_last_error("porla: merged pair ranges need the reduced-radix bucket form"); return PORLA_ERR_STATE; }
        if (after) PORLA_HIP(hipStreamWaitEvent(stream, after, 0));
    }
    if (front_split) PORLA_HIP(hipStreamWaitEvent(stream, ws->front_join_ev, 0));
    {
        ProfScope ps("bucket_sum", stream, true);
        hipLaunchKernelGGL((k_bucket_sum30<C>), dim3((unsigned)((max_items + 255) / 256)), dim3(256), 0, stream, pts,
                           (const uint32_t*)ws->entries.p, (const uint32_t*)ws->starts.p,
                           (const uint32_t*)ws->counts.p, (const uint2*)ws->order.p, (const uint32_t*)ws->fill.p,
                           (const uint32_t*)ctrl, bk, (XYZZ<M>*)ws->chunk_out.p, accumulate ? 1u : 0u);
    }
    {
        ProfScope ps("bucket_combine", stream);
        hipLaunchKernelGGL((k_bucket_combine<C>), dim3(2048), dim3(64), 0, stream, (const uint32_t*)ws->heavy.p,
                           (const uint32_t*)ws->fill.p, (const uint32_t*)ws->counts.p, (const uint32_t*)ctrl,
                           (const XYZZ<M>*)ws->chunk_out.p, bk, accumulate ? 1u : 0u);
    }
    PORLA_HIP(hipGetLastError());
    if (forced) return PORLA_OK;
    return msm_tree_launch<C>(ws, bk, c, W, glv, stream);
}

// bucket reduction of W windows of 2^(c-1) buckets each (the bit-sliced tree of msm.hip.h) into the workspace's pinned buffer;
// records ws->done and leaves the shape for msm_finish
template <class C>
static int msm_tree_launch(Workspace* ws, const XYZZ<typename C::Fp>* buckets, int c, int W, bool glv, hipStream_t stream) {
    using M = typename C::Fp;
    int rc;
    const uint32_t B = 1u << (c - 1);
    const size_t nb = (size_t)W * B;
    const uint32_t nlev = (uint32_t)(c - 1);
    // (sizes for the two halves of the window set, each with its own end slack)
    if ((rc = ws->tree_s.ensure((nb + 2) * sizeof(XYZZ<M>)))) return rc;               // all S levels: nb/2 + nb/4 + ...
    if ((rc = ws->tree_m.ensure((2 * (nb / 4) + 6) * sizeof(XYZZ<M>)))) return rc;     // two ping-pong halves
    if ((rc = ws->tree_mt.ensure((2 * (size_t)W * (B / 4 + 1) + 4) * sizeof(XYZZ<M>)))) return rc;  // the tail's private halves
    if (ws->h_windows_cap < (size_t)W * c * sizeof(XYZZ<M>) || ws->h_windows_cap < 64 * 1024) {
        if (ws->h_windows) PORLA_HIP(hipHostFree(ws->h_windows));
        ws->h_windows_cap = 64 * 1024;
        PORLA_HIP(hipHostMalloc(&ws->h_windows, ws->h_windows_cap, hipHostMallocMapped | hipHostMallocCoherent));
    }
    void* h_windows_dev = nullptr;
    PORLA_HIP(hipHostGetDevicePointer(&h_windows_dev, ws->h_windows, 0));
    // The windows are independent trees.  With enough buckets they run as TWO halves on two streams: the first levels of a tree
    // are throughput bound, the last ones (quad levels, the tail) latency bound -- side by side the latency-bound levels of one
    // half run under the large levels of the other.
    // (only for a caller that waits for this one device-resident MSM: -1.7 % at 2^18 and 2^20 pairs; with a second MSM in
    // flight -- the two-phase API -- the other MSM already fills those gaps and the extra stream costs 5 % of the pipelined rate)
    const int parts = (ws->lone && W >= 4 && nb >= ((size_t)1 << 18)) ? 2 : 1;
    if (parts == 2) {
        if (!ws->aux_stream) PORLA_HIP(hipStreamCreateWithFlags(&ws->aux_stream, hipStreamNonBlocking));
        if (!ws->fork_ev) PORLA_HIP(hipEventCreateWithFlags(&ws->fork_ev, hipEventDisableTiming));
        if (!ws->join_ev) PORLA_HIP(hipEventCreateWithFlags(&ws->join_ev, hipEventDisableTiming));
        PORLA_HIP(hipEventRecord(ws->fork_ev, stream));
        PORLA_HIP(hipStreamWaitEvent(ws->aux_stream, ws->fork_ev, 0));
    }
    size_t s_off = 0, m_off = 0, mt_off = 0;
    for (int part = 0, w0 = 0; part < parts; part++) {
        const int Wp = parts == 1 ? W : (part == 0 ? W / 2 : W - W / 2);
        hipStream_t st = part == 0 ? stream : ws->aux_stream;
        const size_t nbp = (size_t)Wp * B;
        const XYZZ<M>* bk = buckets + (size_t)w0 * B;
        // S level l at s_base + (nbp - (nbp >> l)) (nbp/2 + ... + nbp/2^l entries before it); M slots of level l in half l & 1
        XYZZ<M>* s_base = (XYZZ<M>*)ws->tree_s.p + s_off;
        XYZZ<M>* m_half[2] = {(XYZZ<M>*)ws->tree_m.p + m_off, (XYZZ<M>*)ws->tree_m.p + m_off + nbp / 4 + 1};
        XYZZ<M>* mt_base = (XYZZ<M>*)ws->tree_mt.p + mt_off;
        auto s_level = [&](uint32_t l) { return s_base + (nbp - (nbp >> l)); };
        const bool quad = C::F30_LAZY && tree_quad();
        const uint32_t l0 = tree_tail_start(B, nlev, quad);
        {
            ProfScope ps("tree_levels", st);
            for (uint32_t l = 0; l < l0; l++) {
                TreeLevelArgs<M> a;
                a.s_prev = l ? s_level(l - 1) : bk;
                a.s_prev2 = l >= 2 ? s_level(l - 2) : bk;
                a.m_prev = m_half[(l + 1) & 1];
                a.s_out = s_level(l);
                a.m_out = m_half[l & 1];
                a.fin = nullptr;   // only the tail holds the last level
                a.n = (uint32_t)(nbp >> (l + 1));
                a.m_prev_stride = 2 * a.n; a.m_out_stride = a.n;
                a.l = l; a.nlev = nlev; a.last = 0;
                const size_t tasks = (size_t)(l + 1) * a.n;
                if constexpr (C::F30_LAZY) {
                    // a caller that waits for this MSM alone has idle lanes to spend on latency: more levels on four lanes per
                    // addition; with another MSM in flight the extra lane work would cost throughput
                    if (quad && tasks <= (ws->lone ? (size_t)TREE_QUAD_MAX_TASKS_LONE : (size_t)TREE_QUAD_MAX_TASKS)) {
                        hipLaunchKernelGGL((k_tree_level_quad<C>), dim3((unsigned)((4 * tasks + 255) / 256)), dim3(256), 0, st, a);
                        continue;
                    }
                }
                hipLaunchKernelGGL((k_tree_level<C>), dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, st, a);
            }
        }
        {
            ProfScope ps("tree_tail", st);
            TreeTailArgs<M> t;
            t.buckets = bk;
            for (uint32_t l = 0; l < 24; l++) t.s_lev[l] = l < nlev ? s_level(l) : nullptr;
            t.m_global = m_half[(l0 + 1) & 1];
            t.per_window = B / 4 + 1;
            t.m_tail[0] = mt_base; t.m_tail[1] = mt_base + (size_t)Wp * t.per_window + 1;
            t.fin = (XYZZ<M>*)h_windows_dev + (size_t)w0 * c;
            t.nb = (uint32_t)nbp; t.B = B; t.l0 = l0; t.nlev = nlev;
            uint32_t need = (l0 + 2) * (B >> (l0 + 1)) * (quad ? 4u : 1u);
            uint32_t threads = (need + 63) / 64 * 64;
            if (threads > tree_tail_threads()) threads = tree_tail_threads();
            if (threads < 64) threads = 64;
            hipLaunchKernelGGL((k_tree_tail<C, true>), dim3(Wp), dim3(threads), 0, st, t);
        }
        // no copy packet: the last level stores its W * c results straight into the pinned host buffer (a D2H hipMemcpyAsync
        // was seen to block the launching thread for milliseconds while another MSM is in flight)
        PORLA_HIP(hipGetLastError());
        s_off += nbp + 1; m_off += 2 * (nbp / 4) + 2; mt_off += 2 * (size_t)Wp * (B / 4 + 1) + 2;
        w0 += Wp;
    }
    if (parts == 2) {
        PORLA_HIP(hipEventRecord(ws->join_ev, ws->aux_stream));
        PORLA_HIP(hipStreamWaitEvent(stream, ws->join_ev, 0));
    }
    if (!ws->done) PORLA_HIP(hipEventCreateWithFlags(&ws->done, hipEventDisableTiming));
    PORLA_HIP(hipEventRecord(ws->done, stream));
    ws->pend_W = W;
    ws->pend_c = c;
    g_last_shape[0] = c; g_last_shape[1] = W; g_last_shape[2] = glv ? 1 : 0;
    return PORLA_OK;
}

// waits for the launched MSM of this workspace and folds the tree's per-window S / M_k sums on the host
template <class C>
static int msm_finish(Workspace* ws, XYZZ<typename C::Fp>* total) {
    using M = typename C::Fp;
    if (ws->pend_W == 0) { *total = xyzz_inf<M>(); return PORLA_OK; }
    if (ws->pend_W < 0) {       // single-launch path: the kernel left its shape in front of the sums
        // its last action is the release store of the sequence number into the (coherent) pinned header: poll for it instead
        // of sleeping on the event -- the wake-up costs more than the fold -- and fall back to the event after 2 ms
        const volatile uint32_t* hdr = (const volatile uint32_t*)ws->h_windows;
        const auto t_spin = std::chrono::steady_clock::now();
        bool seen = false;
        for (uint32_t it = 0;; it++) {
            if (__atomic_load_n((const uint32_t*)&hdr[0], __ATOMIC_ACQUIRE) == ws->small_seq) { seen = true; break; }
            __builtin_ia32_pause();
            if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(2)) break;
        }
        if (!seen) PORLA_HIP(hipEventSynchronize(ws->done));
        const int W = (int)hdr[1], c = (int)hdr[2];
        ws->pend_W = 0;
        if (hdr[0] != ws->small_seq || W < 1 || c < 1 || c > SMALL_MAX_C || (size_t)W * c * sizeof(XYZZ<M>) + SMALL_HDR_WORDS * 4 > ws->h_windows_cap) {
            set_last_error("porla: the single-launch MSM left no valid result header");
            return PORLA_ERR_HIP;
        }
        g_last_shape[0] = c; g_last_shape[1] = W; g_last_shape[2] = (int)hdr[3];
        *total = h_fold_tree64<M>((const XYZZ<M>*)((const uint8_t*)ws->h_windows + SMALL_HDR_WORDS * 4), W, c);
        return PORLA_OK;
    }
    PORLA_HIP(hipEventSynchronize(ws->done));
    *total = h_fold_tree64<M>((const XYZZ<M>*)ws->h_windows, ws->pend_W, ws->pend_c);
    ws->pend_W = 0;
    return PORLA_OK;
}

template <class C>
static int msm_core(Workspace* ws, const uint8_t* d_scalars, const uint8_t* d_points_be, size_t n, hipStream_t stream,
                    XYZZ<typename C::Fp>* total) {
    int rc = msm_launch<C>(ws, d_scalars, d_points_be, n, stream);
    if (rc) return rc;
    return msm_finish<C>(ws, total);
}

template <class C>
int msm_device(const uint8_t* d_scalars, const uint8_t* d_points, size_t n, hipStream_t stream,
               XYZZ<typename C::Fp>* total) {
    int rc = ensure_device();
    if (rc) return rc;
    Workspace* ws;
    if ((rc = lease_blocking_slot(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu, std::adopt_lock);
    ws->lone = true;
    return msm_core<C>(ws, d_scalars, d_points, n, stream, total);
}
// ONE scalar array over TWO point arrays (device pointers).  Up to SMALL_MAX_N pairs: one launch for both (msm_small_pair_launch);
// above: the two MSMs through the general path, one after the other on this slot (callers that want them overlapped use the
// two-phase form on two streams).
template <class C>
static int msm_pair_locked(Workspace* ws, const uint8_t* d_scalars, const uint8_t* d_points_a, const uint8_t* d_points_b, size_t n,
                           hipStream_t stream, XYZZ<typename C::Fp>* total_a, XYZZ<typename C::Fp>* total_b, int bits_hint = 0) {
    int rc;
    ws->lone = true;
    if (n <= SMALL_MAX_N && g_small_mode != 0 && g_window_override == 0) {
        ws->pend_W = 0;
        if ((rc = msm_small_pair_launch<C>(ws, d_scalars, d_points_a, d_points_b, n, stream, bits_hint))) return rc;
        if ((rc = msm_small_pair_finish<C>(ws, 0, total_a))) return rc;
        return msm_small_pair_finish<C>(ws, 1, total_b);
    }
    if ((rc = msm_core<C>(ws, d_scalars, d_points_a, n, stream, total_a))) return rc;
    return msm_core<C>(ws, d_scalars, d_points_b, n, stream, total_b);
}
template <class C>
int msm_pair_device(const uint8_t* d_scalars, const uint8_t* d_points_a, const uint8_t* d_points_b, size_t n, hipStream_t stream,
                    XYZZ<typename C::Fp>* total_a, XYZZ<typename C::Fp>* total_b) {
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) { *total_a = xyzz_inf<typename C::Fp>(); *total_b = *total_a; return PORLA_OK; }
    Workspace* ws;
    if ((rc = lease_blocking_slot(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu, std::adopt_lock);
    return msm_pair_locked<C>(ws, d_scalars, d_points_a, d_points_b, n, stream, total_a, total_b);
}
// Two-phase form: several MSMs in flight on different streams, each in its own workspace slot (1 .. MSM_USER_SLOTS-1; slot 0
// belongs to the blocking calls).  begin enqueues all kernels and returns; end waits for that slot and folds on the host.
// A slot belongs to the device that was current at begin: end must be called with the same current device.
template <class C>
int msm_device_begin(int slot, const uint8_t* d_scalars, const uint8_t* d_points, size_t n, hipStream_t stream) {
    int rc = ensure_device();
    if (rc) return rc;
    Workspace* ws;
    if ((rc = get_workspace_slot(slot, &ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    if (ws->begun) { set_last_error("porla: MSM slot still has a pending result (call the matching _end first)"); return PORLA_ERR_STATE; }
    ws->lone = false;
    rc = msm_launch<C>(ws, d_scalars, d_points, n, stream);
    if (rc == PORLA_OK) ws->begun = true;
    return rc;
}
template <class C>
int msm_device_end(int slot, XYZZ<typename C::Fp>* total) {
    Workspace* ws;
    int rc = get_workspace_slot(slot, &ws);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    if (!ws->begun || ws->pair_pending) {
        set_last_error("porla: no MSM was begun on this slot of the current device (end must run with the device of its begin; a pair "
                       "is collected with *_audit_msm_pair_end)");
        return PORLA_ERR_STATE;
    }
    ws->begun = false;
    return msm_finish<C>(ws, total);
}

// one range of host pairs -> this slot's staging buffers -> kernels enqueued on the slot's own stream (ws->mu held)
template <class C>
static int msm_host_launch(Workspace* ws, const uint8_t* scalars, const uint8_t* points, size_t n) {
    int rc;
    ws->pend_W = 0;
    if (n == 0) return PORLA_OK;
    if ((rc = ws->in_scalars.ensure(n * 32))) return rc;
    if ((rc = ws->in_points.ensure(n * 64))) return rc;
    hipStream_t s = ws->own_stream;
    PORLA_HIP(hipMemcpyAsync(ws->in_scalars.p, scalars, n * 32, hipMemcpyHostToDevice, s));
    PORLA_HIP(hipMemcpyAsync(ws->in_points.p, points, n * 64, hipMemcpyHostToDevice, s));
    return msm_launch<C>(ws, (const uint8_t*)ws->in_scalars.p, (const uint8_t*)ws->in_points.p, n, s);
}

// pairs per range below which splitting further (or taking another device) costs more than it hides
static inline size_t msm_multi_min_range() {
    static const size_t v = (size_t)1 << 17;
    return v < 1 ? 1 : v;
}
static inline int msm_multi_pipeline() {   // ranges per device when the caller leaves the shard count to the engine
    static const int v = 4;
    return v < 1 ? 1 : (v > 64 ? 64 : v);
}
extern std::mutex g_multi_mu;
static inline bool msm_multi_shared_buckets() {
    static const bool v = !(getenv("PORLA_MSM_SHARED_BUCKETS") && getenv("PORLA_MSM_SHARED_BUCKETS")[0] == '0');
    return v;
}
template <class C>
int msm_host_multi(const uint8_t* scalars, const uint8_t* points, size_t n, int shards, int devices,
                   XYZZ<typename C::Fp>* total) {
    using M = typename C::Fp;
    int rc = ensure_device();
    if (rc) return rc;
    *total = xyzz_inf<M>();
    if (n == 0) return PORLA_OK;
    int visible = 0, first = 0;
    PORLA_HIP(hipGetDeviceCount(&visible));
    PORLA_HIP(hipGetDevice(&first));
    // devices: the caller's count, else PORLA_MSM_DEVICES, else every visible device that still gets a worthwhile range
    int G = devices;
    if (G <= 0 && getenv("PORLA_MSM_DEVICES")) G = atoi(getenv("PORLA_MSM_DEVICES"));
    if (G <= 0) {
        size_t by_size = n / msm_multi_min_range();
        G = by_size < 1 ? 1 : (by_size > (size_t)visible ? visible : (int)by_size);
    }
    if (G > visible) G = visible;
    if (G < 1) G = 1;
    int S = shards;
    if (S <= 0) {
        size_t per_dev = n / G / msm_multi_min_range();
        size_t pipe = (size_t)msm_multi_pipeline();
        S = G * (int)(per_dev < 1 ? 1 : (per_dev > pipe ? pipe : per_dev));
    }
    if ((size_t)S > n) S = (int)n;
    if (S < G) G = S;
    // Device g owns the contiguous ranges [g S / G, (g+1) S / G) of the S ranges (range s = pairs [s n / S, (s+1) n / S)).
    // A device's R ranges share ONE shape and ONE bucket array: each range runs digits and sort on its own slot and then
    // accumulates into the buckets where the previous range left them; the reduction tree + host fold run once per device -- a
    // tree costs the same whatever the range size, so R of them were most of the work after the last upload
    // (PORLA_MSM_SHARED_BUCKETS=0: one complete MSM per range, as before).
    std::vector<XYZZ<M>> part((size_t)G, xyzz_inf<M>());
    std::vector<int> dev_rc((size_t)G, PORLA_OK);
    std::vector<std::string> dev_err((size_t)G);
    std::lock_guard<std::mutex> lk_multi(g_multi_mu);
    auto worker = [&](int g) {
        int r = PORLA_OK;
        auto fail = [&](int code) { if (!r) { r = code; dev_err[g] = porla_gpu_last_error(); } };
        const int dev = (first + g) % visible;
        if (hipSetDevice(dev) != hipSuccess) { set_last_error("porla: hipSetDevice failed"); fail(PORLA_ERR_HIP); dev_rc[g] = r; return; }
        const int s0 = (int)((long long)g * S / G), s1 = (int)((long long)(g + 1) * S / G), R = s1 - s0;
        std::vector<size_t> bound((size_t)R + 1);
        size_t max_cnt = 0;
        for (int j = 0; j <= R; j++) {
            bound[(size_t)j] = (size_t)((unsigned __int128)(s0 + j) * n / (unsigned)S);
            if (j && bound[(size_t)j] - bound[(size_t)j - 1] > max_cnt) max_cnt = bound[(size_t)j] - bound[(size_t)j - 1];
        }
        constexpr int PIPE = MSM_MULTI_SLOTS;
        Workspace* slot_ws[PIPE] = {nullptr, nullptr, nullptr, nullptr};
        const int used_slots = R < PIPE ? R : PIPE;
        // every slot is looked up (registry lock) BEFORE any slot mutex is taken: porla_gpu_release_msm_workspaces holds the
        // registry lock while it try-locks the slots, so a slot mutex is never held across get_workspace_slot
        Workspace* found[PIPE] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < used_slots && !r; k++) {
            int rr = get_workspace_slot(MSM_MULTI_SLOT0 + k, &found[k]);
            if (rr) { found[k] = nullptr; fail(rr); }
        }
        for (int k = 0; k < used_slots && !r; k++) {
            int rr;
            slot_ws[k] = found[k];
            slot_ws[k]->mu.lock();
            slot_ws[k]->pend_W = 0;
            slot_ws[k]->lone = false;       // (the two-stream tree made the host-buffer call slower: 2.72 -> 2.90 ms at 2^20)
            // staging sized for the largest range before any copy is in flight (growing it later would free a buffer in use)
            if ((rr = slot_ws[k]->in_scalars.ensure(max_cnt * 32)) || (rr = slot_ws[k]->in_points.ensure(max_cnt * 64))) { fail(rr); break; }
            if (!slot_ws[k]->merged && hipEventCreateWithFlags(&slot_ws[k]->merged, hipEventDisableTiming) != hipSuccess) {
                set_last_error("porla: hipEventCreate failed"); fail(PORLA_ERR_HIP); break;
            }
        }
        const bool shared = R >= 2 && msm_multi_shared_buckets();
        MsmShape shape = msm_full_shape<C>(max_cnt);
        const size_t nb = (size_t)shape.W << (shape.c - 1);
        XYZZ<M>* acc = nullptr;
        if (!r && shared) {
            int rr = slot_ws[0]->multi_acc.ensure(nb * sizeof(XYZZ<M>));
            if (rr) fail(rr);
            acc = (XYZZ<M>*)slot_ws[0]->multi_acc.p;
        }
        // (measured and dropped: the uploads on a thread of their own, so that a range's kernels are enqueued while the next range
        // is copied, and ranges of unequal size -- neither moved the 2^20 figure, profiles/r02_g_host_boundary_sweep.txt)
        auto upload = [&](int j) -> int {                // copies of local range j into its slot's staging buffers
            Workspace* ws = slot_ws[j % PIPE];
            const size_t lo = bound[(size_t)j], cnt = bound[(size_t)j + 1] - lo;
            if (cnt == 0) return PORLA_OK;
            PORLA_HIP(hipMemcpyAsync(ws->in_scalars.p, scalars + 32 * lo, cnt * 32, hipMemcpyHostToDevice, ws->own_stream));
            PORLA_HIP(hipMemcpyAsync(ws->in_points.p, points + 64 * lo, cnt * 64, hipMemcpyHostToDevice, ws->own_stream));
            return PORLA_OK;
        };
        Workspace* last_ws = nullptr;                    // the slot whose stream carries the reduction
        int prev_merged = -1;                            // slot of the last range that has been merged (its `merged` event is recorded)
        XYZZ<M> unshared_sum = xyzz_inf<M>();
        for (int j = 0; j < R && !r; j++) {
            const int k = j % PIPE;
            Workspace* ws = slot_ws[k];
            if (!shared && ws->pend_W) {                 // one complete MSM per range: fold the slot's previous range first
                XYZZ<M> t;
                int rr = msm_finish<C>(ws, &t);
                if (rr) { fail(rr); break; }
                xyzz_add<M>(unshared_sum, t);
            }
            {
                int rr = upload(j);
                if (rr) { fail(rr); break; }
            }
            const size_t cnt = bound[(size_t)j + 1] - bound[(size_t)j];
            hipStream_t st = ws->own_stream;
            if (cnt) {
                int rr;
                if (!shared) {
                    rr = msm_launch<C>(ws, (const uint8_t*)ws->in_scalars.p, (const uint8_t*)ws->in_points.p, cnt, st);
                } else {
                    const bool first_range = prev_merged < 0;
                    rr = msm_launch<C>(ws, (const uint8_t*)ws->in_scalars.p, (const uint8_t*)ws->in_points.p, cnt, st, &shape, acc,
                                       !first_range, first_range ? nullptr : slot_ws[prev_merged]->merged);
                    if (!rr && hipEventRecord(ws->merged, st) != hipSuccess) { set_last_error("porla: hipEventRecord failed"); rr = PORLA_ERR_HIP; }
                    if (!rr) { prev_merged = k; last_ws = ws; }
                }
                if (rr) { fail(rr); break; }
            }
        }
        if (!r && shared && last_ws) {
            int rr = msm_tree_launch<C>(last_ws, acc, shape.c, shape.W, shape.glv, last_ws->own_stream);
            if (!rr) rr = msm_finish<C>(last_ws, &part[(size_t)g]);
            if (rr) fail(rr);
        }
        for (int k = 0; k < PIPE; k++) {
            if (!slot_ws[k]) continue;
            if (r) { (void)hipStreamSynchronize(slot_ws[k]->own_stream); slot_ws[k]->pend_W = 0; }   // drain what was enqueued
            else if (!shared && slot_ws[k]->pend_W) {
                XYZZ<M> t;
                int rr = msm_finish<C>(slot_ws[k], &t);
                if (rr) fail(rr); else xyzz_add<M>(unshared_sum, t);
            }
            slot_ws[k]->mu.unlock();
        }
        if (!r && !shared) part[(size_t)g] = unshared_sum;
        dev_rc[g] = r;
    };
    if (G == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int g = 1; g < G; g++) th.emplace_back(worker, g);
        worker(0);
        for (auto& t : th) t.join();
        (void)hipSetDevice(first);
    }
    for (int g = 0; g < G; g++) if (dev_rc[g]) { set_last_error(dev_err[g]); return dev_rc[g]; }
    XYZZ<M> acc = xyzz_inf<M>();
    for (int g = 0; g < G; g++) xyzz_add<M>(acc, part[(size_t)g]);
    *total = acc;
    g_last_multi[0] = S; g_last_multi[1] = G;
    return PORLA_OK;
}

// threshold above which the blocking host-buffer form range-splits its input (upload of one range under the kernels of another)
static inline size_t msm_host_split_min() {
    static const size_t v = getenv("PORLA_MSM_SPLIT_MIN") ? (size_t)atoll(getenv("PORLA_MSM_SPLIT_MIN")) : ((size_t)1 << 18);
    return v;
}
// Devices the IMPLICIT range split of a blocking host-buffer call (compute_multi_exp, the secp256k1 shim) may touch.  One
// process that owns the node -- the unmodified C++ Server -- spreads a large call over every visible GPU (0 = automatic,
// DESIGN.md s5).  A rank of a one-process-per-GPU job sees the other ranks' GPUs too and must stay on its own: the call is
// then range-pipelined on the current device only.  PORLA_MSM_DEVICES overrides both; the explicit *_host_multi entry points
// take the caller's count.
int dist_world_size();   // dist.hip: ranks of the in-library RCCL communicator, 0 without one
static inline int msm_implicit_devices() {
    if (getenv("PORLA_MSM_DEVICES")) return 0;   // msm_host_multi reads it
    const char* ws = getenv("WORLD_SIZE");
    const char* lws = getenv("LOCAL_WORLD_SIZE");
    if ((ws && atoi(ws) > 1) || (lws && atoi(lws) > 1) || dist_world_size() > 1) return 1;
    return 0;
}
template <class C>
int msm_host(const uint8_t* scalars, const uint8_t* points, size_t n, XYZZ<typename C::Fp>* total) {
    int rc = ensure_device();
    if (rc) return rc;
    if (n >= msm_host_split_min()) return msm_host_multi<C>(scalars, points, n, 0, msm_implicit_devices(), total);
    Workspace* ws;
    if ((rc = lease_blocking_slot(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu, std::adopt_lock);
    if (n == 0) { *total = xyzz_inf<typename C::Fp>(); return PORLA_OK; }
    ws->lone = true;
    if ((rc = msm_host_launch<C>(ws, scalars, points, n))) return rc;
    return msm_finish<C>(ws, total);
}

// The audit's pair straight from the server's resident MAC arrays (Server.hpp:838-848 / :893-901 collect, per challenged index, the
// MAC commitment, the alignment MAC and the abs(int32) coefficient into ptc / pta / sc before the two MSMs): the gather runs on the
// device -- scalar i = coef[i] as a 32-byte big-endian integer, points a_i = store_a[idx[i]], b_i = store_b[idx[i]] -- into the slot's
// staging buffers, then the pair MSM on the same stream.
static __global__ void __launch_bounds__(256)
k_audit_gather(const uint8_t* __restrict__ store_a, const uint8_t* __restrict__ store_b, const uint64_t* __restrict__ idx,
               const uint32_t* __restrict__ coef, uint32_t n, uint8_t* __restrict__ scalars, uint8_t* __restrict__ pts_a,
               uint8_t* __restrict__ pts_b) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t >> 2, q = t & 3u;           // four lanes per pair: 16 bytes of each point
    if (i >= n) return;
    const uint64_t src = idx[i];
    reinterpret_cast<uint4*>(pts_a + 64 * (size_t)i)[q] = reinterpret_cast<const uint4*>(store_a + 64 * src)[q];
    reinterpret_cast<uint4*>(pts_b + 64 * (size_t)i)[q] = reinterpret_cast<const uint4*>(store_b + 64 * src)[q];
    if (q < 2) {
        uint4 z = make_uint4(0, 0, 0, 0);
        if (q == 1) z.w = __builtin_bswap32(coef[i]);     // bytes 28..31 of the big-endian scalar
        reinterpret_cast<uint4*>(scalars + 32 * (size_t)i)[q] = z;
    }
}
template <class C>
int msm_pair_gather_device(const uint8_t* d_store_a, const uint8_t* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef, size_t n,
                           hipStream_t stream, XYZZ<typename C::Fp>* total_a, XYZZ<typename C::Fp>* total_b) {
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) { *total_a = xyzz_inf<typename C::Fp>(); *total_b = *total_a; return PORLA_OK; }
    if (n >= (1ull << 30)) { set_last_error("porla: too many challenged rows for one call"); return PORLA_ERR_ARG; }
    Workspace* ws;
    if ((rc = lease_blocking_slot(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu, std::adopt_lock);
    if ((rc = ws->in_scalars.ensure(n * 32))) return rc;
    if ((rc = ws->in_points.ensure(n * 128))) return rc;
    uint8_t* d_pts = (uint8_t*)ws->in_points.p;
    hipLaunchKernelGGL(k_audit_gather, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, stream, d_store_a, d_store_b, d_idx, d_coef,
                       (uint32_t)n, (uint8_t*)ws->in_scalars.p, d_pts, d_pts + n * 64);
    PORLA_HIP(hipGetLastError());
    return msm_pair_locked<C>(ws, (const uint8_t*)ws->in_scalars.p, d_pts, d_pts + n * 64, n, stream, total_a, total_b, 32);
}

// Two-phase form of the same (slots as msm_device_begin): begin gathers and launches on `stream` and returns, end polls both result
// regions and folds -- the audit's other chain (row combine -> alignment commitment -> proof) runs in between.  Up to SMALL_MAX_N rows.
template <class C>
int msm_pair_gather_begin(int slot, const uint8_t* d_store_a, const uint8_t* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef, size_t n,
                          hipStream_t stream) {
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0 || n > SMALL_MAX_N || g_small_mode == 0 || g_window_override != 0) {
        set_last_error("porla: the two-phase audit pair takes 1 .. 32768 challenged rows (single-launch path on)");
        return PORLA_ERR_ARG;
    }
    Workspace* ws;
    if ((rc = get_workspace_slot(slot, &ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    if (ws->begun) { set_last_error("porla: MSM slot still has a pending result (call the matching _end first)"); return PORLA_ERR_STATE; }
    ws->lone = false;
    if ((rc = ws->in_scalars.ensure(n * 32))) return rc;
    if ((rc = ws->in_points.ensure(n * 128))) return rc;
    uint8_t* d_pts = (uint8_t*)ws->in_points.p;
    hipLaunchKernelGGL(k_audit_gather, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, stream, d_store_a, d_store_b, d_idx, d_coef,
                       (uint32_t)n, (uint8_t*)ws->in_scalars.p, d_pts, d_pts + n * 64);
    PORLA_HIP(hipGetLastError());
    ws->pend_W = 0;
    if ((rc = msm_small_pair_launch<C>(ws, (const uint8_t*)ws->in_scalars.p, d_pts, d_pts + n * 64, n, stream, 32))) return rc;
    ws->begun = true;
    ws->pair_pending = true;
    return PORLA_OK;
}
template <class C>
int msm_pair_end(int slot, XYZZ<typename C::Fp>* total_a, XYZZ<typename C::Fp>* total_b) {
    Workspace* ws;
    int rc = get_workspace_slot(slot, &ws);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    if (!ws->begun || !ws->pair_pending) {
        set_last_error("porla: no MSM pair was begun on this slot of the current device");
        return PORLA_ERR_STATE;
    }
    ws->begun = false;
    ws->pair_pending = false;
    if ((rc = msm_small_pair_finish<C>(ws, 0, total_a))) return rc;
    return msm_small_pair_finish<C>(ws, 1, total_b);
}

// host-pointer form of the pair: one upload of the scalars, both point arrays side by side in the slot's staging buffer
template <class C>
int msm_pair_host(const uint8_t* scalars, const uint8_t* points_a, const uint8_t* points_b, size_t n, XYZZ<typename C::Fp>* total_a,
                  XYZZ<typename C::Fp>* total_b) {
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) { *total_a = xyzz_inf<typename C::Fp>(); *total_b = *total_a; return PORLA_OK; }
    if (n >= msm_host_split_min()) {
        if ((rc = msm_host<C>(scalars, points_a, n, total_a))) return rc;
        return msm_host<C>(scalars, points_b, n, total_b);
    }
    Workspace* ws;
    if ((rc = lease_blocking_slot(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu, std::adopt_lock);
    if ((rc = ws->in_scalars.ensure(n * 32))) return rc;
    if ((rc = ws->in_points.ensure(n * 128))) return rc;
    hipStream_t s = ws->own_stream;
    uint8_t* d_pts = (uint8_t*)ws->in_points.p;
    PORLA_HIP(hipMemcpyAsync(ws->in_scalars.p, scalars, n * 32, hipMemcpyHostToDevice, s));
    PORLA_HIP(hipMemcpyAsync(d_pts, points_a, n * 64, hipMemcpyHostToDevice, s));
    PORLA_HIP(hipMemcpyAsync(d_pts + n * 64, points_b, n * 64, hipMemcpyHostToDevice, s));
    return msm_pair_locked<C>(ws, (const uint8_t*)ws->in_scalars.p, d_pts, d_pts + n * 64, n, s, total_a, total_b);
}

}  // namespace porla
