// Host-side launch logic of the batched fixed-base commitment (fixed_base.hip.h); included by the per-curve TUs.
#pragma once
#include "engine.hpp"
#include "fixed_base.hip.h"
#include "host_fold64.hpp"
#include <cstdlib>
#include <chrono>
#include <cstring>

namespace porla {

template <class C>
void FixedBase<C>::release() {
    if (table) (void)hipFree(table);
    if (pow_buf) (void)hipFree(pow_buf);
    if (scratch_buf) (void)hipFree(scratch_buf);
    pow_buf = nullptr; scratch_buf = nullptr; table_cap = pow_cap = scratch_cap = 0;
    if (partial) (void)hipFree(partial);
    if (io_rows) (void)hipFree(io_rows);
    if (io_out) (void)hipFree(io_out);
    if (io_stream2) (void)hipStreamDestroy(io_stream2);
    io_stream2 = nullptr;
    table = nullptr; partial = nullptr; io_rows = nullptr; io_out = nullptr;
    partial_cap = io_rows_cap = io_out_cap = 0;
    n_points = 0;
    fence.reset();
    if (h_small) (void)hipHostFree(h_small);
    if (d_small) (void)hipFree(d_small);
    if (small_done) (void)hipEventDestroy(small_done);
    h_small = nullptr; d_small = nullptr; small_done = nullptr;
}

// Builds the multiples table for `n` base points (Montgomery affine, device memory).
template <class C>
int FixedBase<C>::build(const Affine<typename C::Fp>* d_base, size_t n, int window_bits, hipStream_t stream) {
    using M = typename C::Fp;
    n_points = 0;
    if (n == 0) return PORLA_OK;
    // window_bits = 0 -> automatic: the widest window (<= 20 bits) whose table fits min(a quarter of the free HBM, the cap), then
    // the narrowest one with the same window count.  The cap is PORLA_COMMIT_TABLE_GB GiB when set, else A FIFTH OF THE DEVICE'S
    // HBM: on the 288 GB of an MI355X that admits the 20-bit table of the 128-point SRS (52 GiB: 13 additions per coefficient
    // instead of the 15 of the 17-bit, 7.5 GiB table; same box: 7.66 -> 8.58 M commitments/s, build 0.07 -> 0.33 s,
    // profiles/r05_e_commit_table_budget.txt) -- the engine is laid out for this part's memory, and a caller that wants the HBM
    // for something else says so (PORLA_COMMIT_TABLE_GB=16 gives the round-4 table back).
    int cc = window_bits;
    const bool automatic = cc <= 0;
    const char* env_w = getenv("PORLA_COMMIT_WINDOW");
    const bool env_set = env_w && atoi(env_w) > 0;          // (PORLA_COMMIT_WINDOW=0 means what leaving it unset means: automatic)
    if (automatic) cc = env_set ? atoi(env_w) : 20;
    if (cc < 2) cc = 2;
    if (cc > 20) cc = 20;
    PORLA_HIP(hipGetDevice(&device));
    {   // commits that still read the old table: wait for them before it is rebuilt (or freed: hipFree synchronises anyway)
        int rcf = fence.enter(stream);
        if (rcf) return rcf;
    }
    size_t free_b = 0, total_b = 0;
    PORLA_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += table_cap;
    size_t budget = free_b / 4;
    if (automatic && !env_set) {
        const char* g = getenv("PORLA_COMMIT_TABLE_GB");
        const size_t cap = g ? (size_t)(atof(g) * 1073741824.0) : total_b / 5;
        if (cap < budget) budget = cap;
    }
    for (;; cc--) {  // shrink the window until the table fits the budget
        int Wc = (C::SCALAR_BITS + 1 + cc - 1) / cc;
        size_t bytes = n * (size_t)Wc * ((size_t)1 << (cc - 1)) * sizeof(Affine<M>);
        if (bytes <= budget || cc <= 4) break;
    }
    // a narrower window that needs no more windows per coefficient halves the table for the same number of additions
    // (BN254: 255 signed bits are 15 windows of 18 bits -- and 15 of 17)
    if (automatic && !env_set)
        while (cc > 4 && (C::SCALAR_BITS + 1 + (cc - 1) - 1) / (cc - 1) == (C::SCALAR_BITS + 1 + cc - 1) / cc) cc--;
    c = cc;
    W = (C::SCALAR_BITS + 1 + c - 1) / c;
    const uint32_t H = 1u << (c - 1);
    const size_t pairs = n * (size_t)W;
    const size_t entries = pairs * H;
    // scratch: at most 1 GiB of XYZZ entries per batch of (point, window) pairs
    size_t batch_pairs = ((size_t)1 << 30) / ((size_t)H * sizeof(XYZZ<M>));
    if (batch_pairs < 1) batch_pairs = 1;
    if (batch_pairs > pairs) batch_pairs = pairs;
    auto ensure = [](void** ptr, size_t* cap, size_t bytes) -> hipError_t {
        if (bytes <= *cap) return hipSuccess;
        if (*ptr) (void)hipFree(*ptr);
        *ptr = nullptr; *cap = 0;
        hipError_t e = hipMalloc(ptr, bytes);
        if (e == hipSuccess) *cap = bytes;
        return e;
    };
    hipError_t e;
    if ((e = ensure((void**)&table, &table_cap, entries * sizeof(Affine<M>))) != hipSuccess) return hip_fail(e, "hipMalloc(table)", __FILE__, __LINE__);
    if ((e = ensure((void**)&pow_buf, &pow_cap, pairs * sizeof(XYZZ<M>))) != hipSuccess) return hip_fail(e, "hipMalloc(pow)", __FILE__, __LINE__);
    if ((e = ensure((void**)&scratch_buf, &scratch_cap, batch_pairs * H * sizeof(XYZZ<M>))) != hipSuccess) return hip_fail(e, "hipMalloc(scratch)", __FILE__, __LINE__);
    XYZZ<M>* pow = pow_buf;
    XYZZ<M>* scratch = scratch_buf;
    {
        ProfScope ps("fb_base_powers", stream);
        hipLaunchKernelGGL((k_fb_base_powers<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_base, (uint32_t)n, c, W, pow);
    }
    // lane groups of k_fb_multiples: 16 entries per lane for small tables, 64 lanes x up to 64 entries for large ones
    const uint32_t lanes = H >= 1024 ? 64 : (H >= 16 ? H / 16 : 1);
    uint32_t K = H / lanes;
    if (K > 64) K = 64;
    const uint32_t runs = H / (lanes * K);
    const uint32_t groups_per_wave = 64 / lanes;
    for (size_t p0 = 0; p0 < pairs; p0 += batch_pairs) {
        size_t np = pairs - p0 < batch_pairs ? pairs - p0 : batch_pairs;
        {
            ProfScope ps("fb_multiples", stream);
            hipLaunchKernelGGL((k_fb_multiples<C>), dim3((unsigned)((np * runs + groups_per_wave - 1) / groups_per_wave)), dim3(64), 0,
                               stream, (const XYZZ<M>*)pow, (uint32_t)p0, (uint32_t)np, H, lanes, K, scratch);
        }
        size_t cnt = np * H;
        {
            ProfScope ps("fb_normalize", stream);
            hipLaunchKernelGGL((k_fb_normalize<C>), dim3((unsigned)((cnt + 64 * 32 - 1) / (64 * 32))), dim3(64), 0, stream,
                               (const XYZZ<M>*)scratch, cnt, table + p0 * H);
        }
    }
    if (!keep_build_buffers) {
        hipError_t e2 = hipStreamSynchronize(stream);
        (void)hipFree(scratch_buf); (void)hipFree(pow_buf);
        scratch_buf = nullptr; pow_buf = nullptr; scratch_cap = pow_cap = 0;
        if (e2 != hipSuccess) return hip_fail(e2, "fixed-base table construction", __FILE__, __LINE__);
    }
    PORLA_HIP(hipGetLastError());
    n_points = n;
    return fence.leave(stream);
}

template <class C>
int FixedBase<C>::commit_device(const uint8_t* d_rows, size_t n_rows, size_t n_coeffs, size_t row_stride, uint8_t* d_out,
                                hipStream_t stream, bool guest_room) {
    using M = typename C::Fp;
    if (n_rows == 0) return PORLA_OK;
    if (!table || n_coeffs > n_points) { set_last_error("porla: fixed base not built / too few base points"); return PORLA_ERR_STATE; }
    if (n_rows > 0xfffffff0u || n_coeffs > 0xffffu) { set_last_error("porla: commit batch too large"); return PORLA_ERR_ARG; }
    if (n_coeffs == 0) { if (d_out) PORLA_HIP(hipMemsetAsync(d_out, 0, n_rows * 64, stream)); return PORLA_OK; }
    {
        int cur = -1;
        PORLA_HIP(hipGetDevice(&cur));
        if (cur != device) { set_last_error("porla: this fixed-base table lives on another device than the current one"); return PORLA_ERR_STATE; }
        int rcf = fence.enter(stream);
        if (rcf) return rcf;
    }
    // slices per row: enough lanes for TWO to three rounds of 256 CUs x 4 SIMDs x 3 waves -- with 196 608 (one round and a third:
    // 1024 blocks where 768 are resident) the last third ran at one wave per SIMD: 18.1-18.3 ms for 2^17 rows, 17.8 with 393 216
    static const size_t target = 393216;
    // ... but not below 16 additions per lane once every SIMD has two waves anyway (131 072 rows): the fold of a slice pair costs
    // what the shorter chains save (the two-coefficient rows of the MAC batch: 0.455 -> 0.437 ms at 131 072 rows, no difference
    // at 65 536, and below that the slices win)
    uint32_t S = 1;
    while ((size_t)n_rows * S < target && S * 2 <= n_coeffs && S < 128 &&
           !(n_rows >= 131072 && (size_t)(n_coeffs / (2 * S)) * (size_t)W < 16)) S *= 2;
    uint32_t G = S < 64 ? S : 64;
    size_t need = n_rows * (size_t)S * sizeof(XYZZ<M>);
    if (need > partial_cap) {
        if (partial) PORLA_HIP(hipFree(partial));
        partial = nullptr; partial_cap = 0;
        PORLA_HIP(hipMalloc((void**)&partial, need + need / 8));
        partial_cap = need + need / 8;
    }
    {
        ProfScope ps("fb_commit", stream, true);
        if (guest_room)
            hipLaunchKernelGGL((k_fb_commit<C, true>), dim3((unsigned)((n_rows + 255) / 256), S), dim3(256), 0, stream, d_rows,
                               (uint32_t)n_rows, (uint32_t)n_coeffs, row_stride, (const Affine<M>*)table, c, W, S, partial);
        else
            hipLaunchKernelGGL((k_fb_commit<C>), dim3((unsigned)((n_rows + 255) / 256), S), dim3(256), 0, stream, d_rows,
                               (uint32_t)n_rows, (uint32_t)n_coeffs, row_stride, (const Affine<M>*)table, c, W, S, partial);
    }
    if (S > 1) {
        ProfScope ps("fb_fold", stream);
        if constexpr (C::F30_BUCKETS && C::F30_LAZY) {
            const uint32_t rows_per_block = 128 / S;          // S / 2 quads per row, 64 quads per block
            hipLaunchKernelGGL((k_fb_fold_quad<C>), dim3((unsigned)((n_rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, stream,
                               partial, (uint32_t)n_rows, S);
        } else {
            uint32_t rows_per_wave = 64 / G;
            hipLaunchKernelGGL((k_fb_fold<C>), dim3((unsigned)((n_rows + rows_per_wave - 1) / rows_per_wave)), dim3(64), 0,
                               stream, partial, (uint32_t)n_rows, S, G);
        }
    }
    last_S = S;
    if (d_out) {
        ProfScope ps("fb_finish", stream);
        static const size_t spread = FB_FINISH_SPREAD;
        if (n_rows <= spread)
            hipLaunchKernelGGL((k_fb_finish<C, 1>), dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, stream,
                               (const XYZZ<M>*)partial, (uint32_t)n_rows, S, d_out);
        else if (n_rows <= 2 * spread)
            hipLaunchKernelGGL((k_fb_finish<C, 2>), dim3((unsigned)((n_rows + 127) / 128)), dim3(64), 0, stream,
                               (const XYZZ<M>*)partial, (uint32_t)n_rows, S, d_out);
        else
            hipLaunchKernelGGL((k_fb_finish<C, FB_FINISH_ROWS_MAX>),
                               dim3((unsigned)((n_rows + 64 * FB_FINISH_ROWS_MAX - 1) / (64 * FB_FINISH_ROWS_MAX))), dim3(64), 0, stream,
                               (const XYZZ<M>*)partial, (uint32_t)n_rows, S, d_out);
    }
    PORLA_HIP(hipGetLastError());
    return fence.leave(stream);
}

// ---- the single-launch path for a handful of rows
constexpr size_t FB_SMALL_ROW_BYTES = 8192;          // rows of up to 256 coefficients
// staging layout: header | row sums (one 128-byte XYZZ per row) | rows, the rows on a 4 KB boundary past the sums
constexpr size_t FB_SMALL_HDR = 128, FB_SMALL_SUMS = FB_SMALL_HDR, FB_SMALL_SUM_BYTES = 128;
constexpr size_t FB_SMALL_ROWS = (FB_SMALL_SUMS + (size_t)FB_SMALL_MAX_ROWS * FB_SMALL_SUM_BYTES + 4095) / 4096 * 4096;
static inline bool fb_small_enabled() {
    static const bool on = !(getenv("PORLA_COMMIT_SMALL") && getenv("PORLA_COMMIT_SMALL")[0] == '0');
    return on;
}
template <class C>
bool FixedBase<C>::small_ok(size_t n_rows, size_t n_coeffs) {
    return fb_small_enabled() && C::F30_BUCKETS && n_rows >= 1 && n_rows <= (size_t)FB_SMALL_MAX_ROWS && n_coeffs >= 1 &&
           n_coeffs * 32 <= FB_SMALL_ROW_BYTES;
}
template <class C>
int FixedBase<C>::commit_small(const uint8_t* const* row_ptrs, size_t n_rows, size_t n_coeffs, uint8_t* const* outs,
                               hipStream_t stream, const uint8_t* d_rows, XYZZ<typename C::Fp>* raw_sums) {
    using M = typename C::Fp;
    if (!table || n_coeffs > n_points) { set_last_error("porla: fixed base not built / too few base points"); return PORLA_ERR_STATE; }
    int cur = -1;
    PORLA_HIP(hipGetDevice(&cur));
    if (cur != device) { set_last_error("porla: this fixed-base table lives on another device than the current one"); return PORLA_ERR_STATE; }
    static_assert(sizeof(XYZZ<M>) == FB_SMALL_SUM_BYTES, "staging layout: the row sums must end before the rows begin");
    const size_t part_bytes = (size_t)FB_SMALL_MAX_ROWS * FB_SMALL_MAX_SLICES * sizeof(XYZZ<M>);
    if (!h_small) {
        PORLA_HIP(hipHostMalloc(&h_small, FB_SMALL_ROWS + FB_SMALL_MAX_ROWS * FB_SMALL_ROW_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
        PORLA_HIP(hipMalloc(&d_small, part_bytes + 1024));
        PORLA_HIP(hipMemsetAsync(d_small, 0, part_bytes + 1024, stream));
        PORLA_HIP(hipEventCreateWithFlags(&small_done, hipEventDisableTiming));
    }
    uint8_t* hs = (uint8_t*)h_small;
    const size_t stride = n_coeffs * 32;
    if (!d_rows)
        for (size_t r = 0; r < n_rows; r++) memcpy(hs + FB_SMALL_ROWS + r * stride, row_ptrs[r], stride);
    void* h_dev = nullptr;
    PORLA_HIP(hipHostGetDevicePointer(&h_dev, h_small, 0));
    if (++small_seq == 0) small_seq = 1;
    volatile uint32_t* hdr = (volatile uint32_t*)hs;
    hdr[0] = 0;
    const uint32_t P = (uint32_t)(n_coeffs * (size_t)W);
    constexpr uint32_t per_slice = 256;      // (coefficient, window) pairs per slice (512: one row 0.0615 ms, the audit 0.150 ms; 256: 0.060 / 0.142)
    uint32_t SL = (P + per_slice - 1) / per_slice;
    if (SL < 1) SL = 1;
    if (SL > (uint32_t)FB_SMALL_MAX_SLICES) SL = FB_SMALL_MAX_SLICES;
    int rc = fence.enter(stream);
    if (rc) return rc;
    {
        ProfScope ps("fb_commit_small", stream, true);
        hipLaunchKernelGGL((k_fb_commit_small<C>), dim3((unsigned)(n_rows * SL)), dim3(SMALL_THREADS), 0, stream,
                           d_rows ? d_rows : (const uint8_t*)h_dev + FB_SMALL_ROWS, (uint32_t)n_rows, (uint32_t)n_coeffs, stride,
                           (const Affine<M>*)table, c, W, SL,
                           (XYZZ<M>*)d_small, (uint32_t*)((uint8_t*)d_small + part_bytes), (uint32_t*)h_dev,
                           (XYZZ<M>*)((uint8_t*)h_dev + FB_SMALL_SUMS), small_seq);
    }
    PORLA_HIP(hipGetLastError());
    PORLA_HIP(hipEventRecord(small_done, stream));
    if ((rc = fence.leave(stream))) return rc;
    const auto t_spin = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint32_t it = 0;; it++) {
        if (__atomic_load_n((const uint32_t*)&hdr[0], __ATOMIC_ACQUIRE) == small_seq) { seen = true; break; }
        __builtin_ia32_pause();
        if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(2)) break;
    }
    if (!seen) {
        PORLA_HIP(hipEventSynchronize(small_done));
        if (hdr[0] != small_seq) { set_last_error("porla: the single-launch commitment left no result"); return PORLA_ERR_HIP; }
    }
    const XYZZ<M>* sums = (const XYZZ<M>*)(hs + FB_SMALL_SUMS);
    if (raw_sums) {
        for (size_t r = 0; r < n_rows; r++) raw_sums[r] = sums[r];
        return PORLA_OK;
    }
    Affine<M> aff[FB_SMALL_MAX_ROWS];
    h_batch_xyzz_to_affine64<M>(sums, n_rows, aff);       // one inversion for the whole batch
    for (size_t r = 0; r < n_rows; r++) h_affine_to_bytes<M>(outs[r], aff[r]);
    return PORLA_OK;
}

template <class C>
int FixedBase<C>::commit_host(const uint8_t* rows, size_t n_rows, size_t n_coeffs, size_t row_stride, uint8_t* out,
                              hipStream_t stream) {
    if (n_rows == 0) return PORLA_OK;
    if (small_ok(n_rows, n_coeffs)) {
        const uint8_t* rp[FB_SMALL_MAX_ROWS];
        uint8_t* op[FB_SMALL_MAX_ROWS];
        for (size_t r = 0; r < n_rows; r++) { rp[r] = rows + r * row_stride; op[r] = out + 64 * r; }
        return commit_small(rp, n_rows, n_coeffs, op, stream);
    }
    // Large batches go through the staging buffers in chunks of 64 MiB of rows, two staging halves on two streams: chunk k + 1 is
    // copied in while chunk k is committed (the commits follow each other through `fence`), and chunk k's results are copied out
    // only after chunk k + 1 has been enqueued -- a pageable device-to-host copy holds the host thread until the data is there.
    // One copy of 512 MiB, then one commit, then one copy back took 48-51 ms for 2^17 rows (tools/bench_commit_host.py).
    static const size_t chunk_bytes = (size_t)64 << 20;
    size_t chunk_rows = row_stride ? chunk_bytes / row_stride : n_rows;
    if (chunk_rows < HOST_FINISH_MAX_ROWS + 1) chunk_rows = HOST_FINISH_MAX_ROWS + 1;
    if (chunk_rows > n_rows) chunk_rows = n_rows;
    const size_t n_chunks = (n_rows + chunk_rows - 1) / chunk_rows;
    const size_t halves = n_chunks > 1 ? 2 : 1;
    const size_t in_bytes = (chunk_rows - 1) * row_stride + n_coeffs * 32;
    const size_t in_half = (in_bytes + 255) & ~(size_t)255, out_half = (chunk_rows * 64 + 255) & ~(size_t)255;
    if (halves * in_half > io_rows_cap) {
        if (io_rows) PORLA_HIP(hipFree(io_rows));
        io_rows = nullptr; io_rows_cap = 0;
        PORLA_HIP(hipMalloc((void**)&io_rows, halves * in_half + 256));
        io_rows_cap = halves * in_half + 256;
    }
    if (halves * out_half > io_out_cap) {
        if (io_out) PORLA_HIP(hipFree(io_out));
        io_out = nullptr; io_out_cap = 0;
        PORLA_HIP(hipMalloc((void**)&io_out, halves * out_half + 256));
        io_out_cap = halves * out_half + 256;
    }
    if (n_rows <= HOST_FINISH_MAX_ROWS && n_coeffs > 0) {
        // up to a few hundred rows: the projective sums come back (one strided copy) and the host normalises them with one
        // inversion per 64 rows (h_batch_xyzz_to_affine64, ~0.3 us per row) -- the device's finish kernel costs 0.05 ms whatever
        // the batch (one division-step inversion on lone waves) plus the output copy: level at ~256 rows
        using M = typename C::Fp;
        if (in_bytes) PORLA_HIP(hipMemcpyAsync(io_rows, rows, in_bytes, hipMemcpyHostToDevice, stream));
        int rc = commit_device(io_rows, n_rows, n_coeffs, row_stride, nullptr, stream);
        if (rc) return rc;
        std::vector<XYZZ<M>> sums(n_rows);
        PORLA_HIP(hipMemcpy2DAsync(sums.data(), sizeof(XYZZ<M>), partial, (size_t)last_S * sizeof(XYZZ<M>), sizeof(XYZZ<M>), n_rows,
                                   hipMemcpyDeviceToHost, stream));
        PORLA_HIP(hipStreamSynchronize(stream));
        std::vector<Affine<M>> aff(n_rows);
        h_batch_xyzz_to_affine64<M>(sums.data(), n_rows, aff.data());
        for (size_t r = 0; r < n_rows; r++) h_affine_to_bytes<M>(out + 64 * r, aff[r]);
        return PORLA_OK;
    }
    if (halves == 2 && !io_stream2) PORLA_HIP(hipStreamCreateWithFlags(&io_stream2, hipStreamNonBlocking));
    hipStream_t st[2] = {stream, halves == 2 ? io_stream2 : stream};
    auto enqueue = [&](size_t k) -> int {
        const size_t lo = k * chunk_rows, m = n_rows - lo < chunk_rows ? n_rows - lo : chunk_rows;
        const size_t bytes = (m - 1) * row_stride + n_coeffs * 32;
        uint8_t* d_in = io_rows + (k & 1) * in_half;
        if (bytes) PORLA_HIP(hipMemcpyAsync(d_in, rows + lo * row_stride, bytes, hipMemcpyHostToDevice, st[k & 1]));
        return commit_device(d_in, m, n_coeffs, row_stride, io_out + (k & 1) * out_half, st[k & 1]);
    };
    int rc = enqueue(0);
    for (size_t k = 0; k < n_chunks && rc == PORLA_OK; k++) {
        if (k + 1 < n_chunks) rc = enqueue(k + 1);
        if (rc != PORLA_OK) break;
        const size_t lo = k * chunk_rows, m = n_rows - lo < chunk_rows ? n_rows - lo : chunk_rows;
        if (hipMemcpyAsync(out + lo * 64, io_out + (k & 1) * out_half, m * 64, hipMemcpyDeviceToHost, st[k & 1]) != hipSuccess) {
            set_last_error("porla: copy of the commitments to the host failed");
            rc = PORLA_ERR_HIP;
        }
    }
    const hipError_t e0 = hipStreamSynchronize(st[0]), e1 = halves == 2 ? hipStreamSynchronize(st[1]) : hipSuccess;
    if (rc == PORLA_OK && (e0 != hipSuccess || e1 != hipSuccess)) { set_last_error("porla: commit batch failed on the device"); rc = PORLA_ERR_HIP; }
    return rc;
}

// base given as 64-byte big-endian affine points on the host (the reference's wire format)
template <class C>
int FixedBase<C>::build_from_host_bytes(const uint8_t* points_be, size_t n, int window_bits, hipStream_t stream) {
    using M = typename C::Fp;
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) { release(); return PORLA_OK; }
    uint8_t* d_in = nullptr;
    Affine<M>* d_mont = nullptr;
    PORLA_HIP(hipMalloc((void**)&d_in, n * 64));
    hipError_t e = hipMalloc((void**)&d_mont, n * sizeof(Affine<M>));
    if (e != hipSuccess) { (void)hipFree(d_in); return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
    (void)hipMemcpyAsync(d_in, points_be, n * 64, hipMemcpyHostToDevice, stream);
    hipLaunchKernelGGL((k_points_to_mont<C, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)d_in,
                       d_mont, (uint32_t)n);
    rc = build(d_mont, n, window_bits, stream);
    (void)hipStreamSynchronize(stream);
    (void)hipFree(d_in);
    (void)hipFree(d_mont);
    return rc;
}

}  // namespace porla
