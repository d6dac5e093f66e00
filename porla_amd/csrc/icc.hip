// ICC encode launch logic + C ABI (include/porla_gpu.h: porla_icc_encode_device / _host).
// Patch site in the reference (no function boundary exists there): Server::CRebuild_Cached,
// porla/Server/Server.hpp:1544-1833; see INTEGRATION.md.
#include "engine.hpp"
#include "icc.cuh"
#include "icc_host.hpp"

#include <cstdlib>
#include <vector>

namespace porla {

struct IccWs {
    int device = -1;
    Buf work, tw, wpow, in, xo, al, sc;
    uint32_t tw_n = 0;
    int tw_curve = -1;
    UseFence fence;   // work / twiddle buffers are shared between calls that may come on different streams
};
static std::mutex g_icc_mu;
static std::vector<IccWs*> g_icc_ws;

static int get_icc_ws(IccWs** out) {
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    for (auto* w : g_icc_ws) if (w->device == dev) { *out = w; return PORLA_OK; }
    IccWs* w = new IccWs();
    w->device = dev;
    g_icc_ws.push_back(w);
    *out = w;
    return PORLA_OK;
}

// twiddle table w^e, e < N (resident across calls with the same N and curve)
template <class Q>
static int ensure_twiddles(IccWs* ws, int curve, size_t n, hipStream_t stream) {
    if (ws->tw_n == n && ws->tw_curve == curve) return PORLA_OK;
    const int logn = ilog2u(n);
    int rc;
    if ((rc = ws->tw.ensure(n * sizeof(IccElem<Q>)))) return rc;
    if ((rc = ws->wpow.ensure(64 * sizeof(Fe<IccFp>)))) return rc;
    std::vector<Fe<IccFp>> wp(logn ? logn : 1);
    Fe<IccFp> cur = icc_root(n);
    for (int i = 0; i < logn; i++) { wp[i] = cur; cur = fe_sqr<IccFp>(cur); }
    PORLA_HIP(hipMemcpyAsync(ws->wpow.p, wp.data(), logn * sizeof(Fe<IccFp>), hipMemcpyHostToDevice, stream));
    PORLA_HIP(hipStreamSynchronize(stream));  // wp is a host temporary
    ProfScope ps("icc_twiddles", stream);
    hipLaunchKernelGGL((k_icc_twiddles<Q>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (IccElem<Q>*)ws->tw.p,
                       (uint32_t)n, (const Fe<IccFp>*)ws->wpow.p, logn);
    ws->tw_n = (uint32_t)n;
    ws->tw_curve = curve;
    return PORLA_OK;
}

template <class Q>
static int icc_mix_core(IccWs* ws, int curve, const uint8_t* d_a0, const uint8_t* d_a1, size_t len, size_t ncols, size_t n_total,
                        uint8_t* d_out, hipStream_t stream) {
    int rc;
    if ((rc = ws->fence.enter(stream))) return rc;
    if ((rc = ensure_twiddles<Q>(ws, curve, n_total, stream))) return rc;
    const size_t total = len * ncols;
    {
        ProfScope ps("icc_mix", stream);
        hipLaunchKernelGGL((k_icc_mix<Q>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_a0, d_a1, (uint32_t)len,
                           (uint32_t)ncols, (const IccElem<Q>*)ws->tw.p, (uint32_t)(n_total / len), d_out);
    }
    PORLA_HIP(hipGetLastError());
    return ws->fence.leave(stream);
}

template <class Q>
static int icc_encode_core(IccWs* ws, int curve, const uint8_t* d_rows, size_t n, size_t ncols, unsigned long long write_step,
                           int part, uint8_t* d_x, uint8_t* d_al, uint8_t* d_sc, int scalar_le, hipStream_t stream,
                           uint8_t* d_qres = nullptr) {
    const int logn = ilog2u(n);
    if (n < 2 || ((size_t)1 << logn) != n || n > (1u << 30) || ncols == 0) {
        set_last_error("porla: ICC encode needs a power-of-two row count >= 2");
        return PORLA_ERR_ARG;
    }
    const size_t total = n * ncols;
    int rc;
    if ((rc = ws->work.ensure(total * sizeof(IccElem<Q>)))) return rc;
    if ((rc = ensure_twiddles<Q>(ws, curve, n, stream))) return rc;
    Fe<IccFp> w = icc_root(n);
    // ---- init scaling: wt = w^reverse_bits(write_step % N, height-1) for the Y part (Server.hpp:1494), 1 for X
    IccElem<Q> wt;
    wt.p = fe_one<IccFp>();
    wt.q = fe_one<Q>();
    int use_wt = 0;
    if (part == 1) {
        const int height = logn + 1;
        uint64_t ex = rev_bits(write_step % n, height - 1);
        uint32_t e[8] = {(uint32_t)ex, (uint32_t)(ex >> 32), 0, 0, 0, 0, 0, 0};
        wt.p = h_fe_pow<IccFp>(w, e);
        Fe<IccFp> plain = fe_from_mont<IccFp>(wt.p);
        Fe<Q> tq;
        for (int k = 0; k < 8; k++) tq.v[k] = plain.v[k];
        fe_reduce_plain<Q>(tq.v, 8);
        wt.q = fe_to_mont<Q>(tq);
        use_wt = 1;
    }
    static const int fused = !(getenv("PORLA_ICC_FUSED") && getenv("PORLA_ICC_FUSED")[0] == '0');
    IccOut out{d_x, d_al, d_sc, d_qres, scalar_le};
    if (fused) {
        // ceil(logn / 8) passes of (almost) equal stage counts, each through LDS tiles of 512 symbols; the first pass reads
        // the raw chunks, the last one writes the outputs: the residue-pair working set only travels between passes
        const int passes = (logn + 7) / 8;
        int s = 1;
        for (int pz = 0; pz < passes; pz++) {
            const int ns = (logn - (s - 1) + (passes - pz) - 1) / (passes - pz);
            int cc_log = ICC_TILE_LOG - ns;                                    // 2^ns rows x 2^cc_log columns = 512 symbols
            while (cc_log > 0 && ((size_t)1 << (cc_log - 1)) >= ncols) cc_log--;   // no wider than the row
            const size_t col_tiles = (ncols + ((size_t)1 << cc_log) - 1) >> cc_log;
            const dim3 grid((unsigned)(col_tiles * (n >> ns)));
            const bool first = pz == 0, last = pz == passes - 1;
            ProfScope ps("icc_fused", stream, true);
#define PORLA_ICC_LAUNCH(F, L)                                                                                              \
    hipLaunchKernelGGL((k_icc_fused<Q, F, L>), grid, dim3(256), 0, stream, (IccElem<Q>*)ws->work.p,                         \
                       (const IccElem<Q>*)ws->tw.p, (uint32_t)n, (uint32_t)ncols, s, ns, cc_log, d_rows, wt, use_wt, out)
            if (first && last) PORLA_ICC_LAUNCH(true, true);
            else if (first) PORLA_ICC_LAUNCH(true, false);
            else if (last) PORLA_ICC_LAUNCH(false, true);
            else PORLA_ICC_LAUNCH(false, false);
#undef PORLA_ICC_LAUNCH
            s += ns;
        }
    } else {
        {
            ProfScope ps("icc_load", stream);
            hipLaunchKernelGGL((k_icc_load<Q>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_rows,
                               (IccElem<Q>*)ws->work.p, total, wt, use_wt);
        }
        int s = 1;
        while (s + 1 <= logn) {
            ProfScope ps("icc_stages_r4", stream);
            size_t groups = (n >> 2) * ncols;
            hipLaunchKernelGGL((k_icc_stages<Q, 2>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream,
                               (IccElem<Q>*)ws->work.p, (const IccElem<Q>*)ws->tw.p, (uint32_t)n, (uint32_t)ncols, s);
            s += 2;
        }
        if (s <= logn) {
            ProfScope ps("icc_stages_r2", stream);
            size_t groups = (n >> 1) * ncols;
            hipLaunchKernelGGL((k_icc_stages<Q, 1>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream,
                               (IccElem<Q>*)ws->work.p, (const IccElem<Q>*)ws->tw.p, (uint32_t)n, (uint32_t)ncols, s);
        }
        {
            ProfScope ps("icc_finish", stream);
            hipLaunchKernelGGL((k_icc_finish<Q>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               (const IccElem<Q>*)ws->work.p, total, out);
        }
    }
    PORLA_HIP(hipGetLastError());
    return PORLA_OK;
}

static int icc_encode_dispatch(IccWs* ws, int curve, const uint8_t* d_rows, size_t n, size_t ncols, unsigned long long ws_step,
                               int part, uint8_t* d_x, uint8_t* d_al, uint8_t* d_sc, int scalar_le, hipStream_t stream,
                               uint8_t* d_qres = nullptr) {
    if (curve != 0 && curve != 1) {
        set_last_error("porla: curve must be 0 (BN254 / KZG) or 1 (secp256k1 / IPA)");
        return PORLA_ERR_ARG;
    }
    int rc = ws->fence.enter(stream);      // an earlier encode on another stream may still use work / the twiddles
    if (rc) return rc;
    rc = curve == 0 ? icc_encode_core<IccBn254Fr>(ws, 0, d_rows, n, ncols, ws_step, part, d_x, d_al, d_sc, scalar_le, stream, d_qres)
                    : icc_encode_core<IccSecp256k1Fn>(ws, 1, d_rows, n, ncols, ws_step, part, d_x, d_al, d_sc, scalar_le, stream, d_qres);
    if (rc) return rc;
    return ws->fence.leave(stream);
}

// The butterfly network as a matrix over Z_q: row k = the coefficients F[k][0..n) with out_k = sum_i F[k][i] * in_i
// (part 1: inputs pre-scaled by wt), 32-byte big-endian each -- the data-side encode applied to the N x N identity.
int icc_network_matrix_device(int curve, size_t n, unsigned long long write_step, int part, uint8_t* d_rows_out,
                              hipStream_t stream) {
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    if ((rc = ws->in.ensure(n * n * 32))) return rc;
    if ((rc = ws->fence.enter(stream))) return rc;
    PORLA_HIP(hipMemsetAsync(ws->in.p, 0, n * n * 32, stream));
    hipLaunchKernelGGL(k_icc_identity, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (uint8_t*)ws->in.p, (uint32_t)n);
    return icc_encode_dispatch(ws, curve, (const uint8_t*)ws->in.p, n, n, write_step, part, nullptr, nullptr, nullptr, 0,
                               stream, d_rows_out);
}

}  // namespace porla

using namespace porla;

extern "C" {

int porla_icc_encode_device(const void* d_rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                            int part, void* d_x_out, void* d_aligned_out, void* d_scalars_out, int scalar_le, void* stream) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!d_rows_in) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    return icc_encode_dispatch(ws, curve, (const uint8_t*)d_rows_in, n_rows, n_cols, write_step, part, (uint8_t*)d_x_out,
                               (uint8_t*)d_aligned_out, (uint8_t*)d_scalars_out, scalar_le, (hipStream_t)stream);
}

int porla_icc_encode_host(const uint8_t* rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                          int part, uint8_t* x_out, uint8_t* aligned_out, uint8_t* scalars_out, int scalar_le) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!rows_in) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    const size_t total = n_rows * n_cols;
    if ((rc = ws->in.ensure(total * 32))) return rc;
    if (x_out && (rc = ws->xo.ensure(total * 64))) return rc;
    if (aligned_out && (rc = ws->al.ensure(total * 32))) return rc;
    if (scalars_out && (rc = ws->sc.ensure(total * 32))) return rc;
    hipStream_t s = nullptr;
    PORLA_HIP(hipMemcpyAsync(ws->in.p, rows_in, total * 32, hipMemcpyHostToDevice, s));
    rc = icc_encode_dispatch(ws, curve, (const uint8_t*)ws->in.p, n_rows, n_cols, write_step, part,
                             x_out ? (uint8_t*)ws->xo.p : nullptr, aligned_out ? (uint8_t*)ws->al.p : nullptr,
                             scalars_out ? (uint8_t*)ws->sc.p : nullptr, scalar_le, s);
    if (rc) return rc;
    if (x_out) PORLA_HIP(hipMemcpyAsync(x_out, ws->xo.p, total * 64, hipMemcpyDeviceToHost, s));
    if (aligned_out) PORLA_HIP(hipMemcpyAsync(aligned_out, ws->al.p, total * 32, hipMemcpyDeviceToHost, s));
    if (scalars_out) PORLA_HIP(hipMemcpyAsync(scalars_out, ws->sc.p, total * 32, hipMemcpyDeviceToHost, s));
    PORLA_HIP(hipStreamSynchronize(s));
    return PORLA_OK;
}

int porla_icc_mix_device(const void* d_a0, const void* d_a1, size_t len, size_t n_cols, size_t n_total, int curve, void* d_out,
                         void* hip_stream) {
    int rc = ensure_device();
    if (rc) return rc;
    const int ll = ilog2u(len), ln = ilog2u(n_total);
    if (!d_a0 || !d_a1 || !d_out || len == 0 || n_cols == 0 || ((size_t)1 << ll) != len || ((size_t)1 << ln) != n_total ||
        len > n_total || n_total < 2 || (curve != 0 && curve != 1)) {
        set_last_error("porla: bad argument to porla_icc_mix_device (len and n_total must be powers of two, len <= n_total)");
        return PORLA_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    if (curve == 0) return icc_mix_core<IccBn254Fr>(ws, 0, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_cols, n_total, (uint8_t*)d_out, (hipStream_t)hip_stream);
    return icc_mix_core<IccSecp256k1Fn>(ws, 1, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_cols, n_total, (uint8_t*)d_out, (hipStream_t)hip_stream);
}

int porla_icc_mix_host(const uint8_t* a0, const uint8_t* a1, size_t len, size_t n_cols, size_t n_total, int curve, uint8_t* out) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!a0 || !a1 || !out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    const size_t bytes = len * n_cols * 64;
    void *d0 = nullptr, *d1 = nullptr, *dout = nullptr;
    PORLA_HIP(hipMalloc(&d0, bytes));
    hipError_t e1 = hipMalloc(&d1, bytes), e2 = hipMalloc(&dout, 2 * bytes);
    if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(dout); return hip_fail(e1 != hipSuccess ? e1 : e2, "hipMalloc", __FILE__, __LINE__); }
    (void)hipMemcpy(d0, a0, bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(d1, a1, bytes, hipMemcpyHostToDevice);
    rc = porla_icc_mix_device(d0, d1, len, n_cols, n_total, curve, dout, nullptr);
    hipError_t e3 = hipMemcpy(out, dout, 2 * bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(dout);
    if (rc) return rc;
    if (e3 != hipSuccess) return hip_fail(e3, "hipMemcpy", __FILE__, __LINE__);
    return PORLA_OK;
}

}  // extern "C"
