// Host-side launch logic of the batched fixed-base commitment (fixed_base.cuh); included by the per-curve TUs.
#pragma once
#include "engine.hpp"
#include "fixed_base.cuh"
#include <cstdlib>

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
    table = nullptr; partial = nullptr; io_rows = nullptr; io_out = nullptr;
    partial_cap = io_rows_cap = io_out_cap = 0;
    n_points = 0;
}

// Builds the multiples table for `n` base points (Montgomery affine, device memory).
template <class C>
int FixedBase<C>::build(const Affine<typename C::Fp>* d_base, size_t n, int window_bits, hipStream_t stream) {
    using M = typename C::Fp;
    n_points = 0;
    if (n == 0) return PORLA_OK;
    // window_bits = 0 -> automatic: the widest window (<= 20 bits) whose table fits min(a quarter of the free HBM,
    // PORLA_COMMIT_TABLE_GB, default 64 GB) -- 20 bits = 56 GB for the 128-point BN254 SRS on a 288 GB MI355X.  Wider is
    // faster (fewer additions per row: 13 windows instead of the 15 of an 18-bit / 16 GB table, 7.4 against 6.65 M
    // commits/s) and the memory is there; the one-off build grows with it (0.43 s against 0.14 s).
    int cc = window_bits;
    const bool automatic = cc <= 0;
    if (automatic) {
        const char* e = getenv("PORLA_COMMIT_WINDOW");
        cc = e ? atoi(e) : 20;
    }
    if (cc < 2) cc = 2;
    if (cc > 20) cc = 20;
    PORLA_HIP(hipGetDevice(&device));
    size_t free_b = 0, total_b = 0;
    PORLA_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += table_cap;
    size_t budget = free_b / 4;
    if (automatic && !getenv("PORLA_COMMIT_WINDOW")) {
        const char* g = getenv("PORLA_COMMIT_TABLE_GB");
        const size_t cap = (size_t)((g ? atof(g) : 64.0) * 1e9);
        if (cap < budget) budget = cap;
    }
    for (;; cc--) {  // shrink the window until the table fits the budget
        int Wc = (C::SCALAR_BITS + 1 + cc - 1) / cc;
        size_t bytes = n * (size_t)Wc * ((size_t)1 << (cc - 1)) * sizeof(Affine<M>);
        if (bytes <= budget || cc <= 4) break;
    }
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
    return PORLA_OK;
}

template <class C>
int FixedBase<C>::commit_device(const uint8_t* d_rows, size_t n_rows, size_t n_coeffs, size_t row_stride, uint8_t* d_out,
                                hipStream_t stream) {
    using M = typename C::Fp;
    if (n_rows == 0) return PORLA_OK;
    if (!table || n_coeffs > n_points) { set_last_error("porla: fixed base not built / too few base points"); return PORLA_ERR_STATE; }
    if (n_rows > 0xfffffff0u || n_coeffs > 0xffffu) { set_last_error("porla: commit batch too large"); return PORLA_ERR_ARG; }
    if (n_coeffs == 0) { if (d_out) PORLA_HIP(hipMemsetAsync(d_out, 0, n_rows * 64, stream)); return PORLA_OK; }
    // slices per row: enough lanes to fill 256 CUs x 4 SIMDs x 3 waves
    static const size_t target = getenv("PORLA_COMMIT_LANES") ? (size_t)atol(getenv("PORLA_COMMIT_LANES")) : (size_t)196608;
    uint32_t S = 1;
    while ((size_t)n_rows * S < target && S * 2 <= n_coeffs && S < 128) S *= 2;
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
        hipLaunchKernelGGL((k_fb_commit<C>), dim3((unsigned)((n_rows + 255) / 256), S), dim3(256), 0, stream, d_rows,
                           (uint32_t)n_rows, (uint32_t)n_coeffs, row_stride, (const Affine<M>*)table, c, W, S, partial);
    }
    if (S > 1) {
        ProfScope ps("fb_fold", stream);
        uint32_t rows_per_wave = 64 / G;
        hipLaunchKernelGGL((k_fb_fold<C>), dim3((unsigned)((n_rows + rows_per_wave - 1) / rows_per_wave)), dim3(64), 0,
                           stream, partial, (uint32_t)n_rows, S, G);
    }
    last_S = S;
    if (d_out) {
        ProfScope ps("fb_finish", stream);
        hipLaunchKernelGGL((k_fb_finish<C>), dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, stream,
                           (const XYZZ<M>*)partial, (uint32_t)n_rows, S, d_out);
    }
    PORLA_HIP(hipGetLastError());
    return PORLA_OK;
}

template <class C>
int FixedBase<C>::commit_host(const uint8_t* rows, size_t n_rows, size_t n_coeffs, size_t row_stride, uint8_t* out,
                              hipStream_t stream) {
    if (n_rows == 0) return PORLA_OK;
    size_t in_bytes = (n_rows - 1) * row_stride + n_coeffs * 32;
    if (in_bytes > io_rows_cap) {
        if (io_rows) PORLA_HIP(hipFree(io_rows));
        io_rows = nullptr; io_rows_cap = 0;
        PORLA_HIP(hipMalloc((void**)&io_rows, in_bytes + 256));
        io_rows_cap = in_bytes + 256;
    }
    if (n_rows * 64 > io_out_cap) {
        if (io_out) PORLA_HIP(hipFree(io_out));
        io_out = nullptr; io_out_cap = 0;
        PORLA_HIP(hipMalloc((void**)&io_out, n_rows * 64 + 256));
        io_out_cap = n_rows * 64 + 256;
    }
    if (in_bytes) PORLA_HIP(hipMemcpyAsync(io_rows, rows, in_bytes, hipMemcpyHostToDevice, stream));
    if (n_rows <= HOST_FINISH_MAX_ROWS && n_coeffs > 0) {
        // a handful of rows (the reference calls compute_digest_from_srs one row at a time): the projective sums come back
        // and the host normalises them -- one inversion costs ~25 us there against ~250 us of dependent products on a lone wave
        using M = typename C::Fp;
        int rc = commit_device(io_rows, n_rows, n_coeffs, row_stride, nullptr, stream);
        if (rc) return rc;
        XYZZ<M> sums[HOST_FINISH_MAX_ROWS];
        for (size_t r = 0; r < n_rows; r++)
            PORLA_HIP(hipMemcpyAsync(&sums[r], partial + r * last_S, sizeof(XYZZ<M>), hipMemcpyDeviceToHost, stream));
        PORLA_HIP(hipStreamSynchronize(stream));
        for (size_t r = 0; r < n_rows; r++) h_affine_to_bytes<M>(out + 64 * r, h_xyzz_to_affine<M>(sums[r]));
        return PORLA_OK;
    }
    int rc = commit_device(io_rows, n_rows, n_coeffs, row_stride, io_out, stream);
    if (rc) return rc;
    PORLA_HIP(hipMemcpyAsync(out, io_out, n_rows * 64, hipMemcpyDeviceToHost, stream));
    PORLA_HIP(hipStreamSynchronize(stream));
    return PORLA_OK;
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
