// ICC encode launch logic + C ABI (include/porla_gpu.h: porla_icc_encode_device / _host).
// Patch site in the reference (no function boundary exists there): Server::CRebuild_Cached,
// porla/Server/Server.hpp:1544-1833; see INTEGRATION.md.
#include "engine.hpp"
#include "icc.hip.h"
#include "icc30.hip.h"
#include "icc30_split.hip.h"
#include "icc_host.hpp"

#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace porla {

struct IccWs {
    int device = -1;
    Buf work, tw, tw30, tw30p, tw30q, wpow, in, xo, al, sc, park_y;
    uint32_t tw_n = 0, tw30_n = 0;   // tw30: the same table in the reduced-radix form of icc30.hip.h (80-byte slots)
    uint32_t tw30s_n = 0;            // tw30p / tw30q: the plane tables of icc30_split.hip.h (40-byte slots)
    int tw30_curve = -1, tw30s_curve = -1;
    int tw_curve = -1;
    UseFence fence;   // work / twiddle buffers are shared between calls that may come on different streams
    std::mutex mu;    // one encode at a time per device (the column-range splitter runs one host thread per device)
};
static std::mutex g_icc_mu;    // the registry only
static std::vector<IccWs*> g_icc_ws;

static int get_icc_ws(IccWs** out) {
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_icc_mu);
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
    ws->tw30_n = 0;
    ws->tw30s_n = 0;
    return PORLA_OK;
}

template <class Q>
static int icc_mix_core(IccWs* ws, int curve, const uint8_t* d_a0, const uint8_t* d_a1, size_t len, size_t ncols, size_t n_total,
                        uint8_t* d_out, hipStream_t stream) {
    int rc;
    if ((rc = ws->fence.enter(stream))) return rc;
    if ((rc = ensure_twiddles<Q>(ws, curve, n_total, stream))) return rc;
    const size_t total = len * ncols;
    // the reduced-radix kernel (icc30.hip.h:k_icc_mix30) and its twiddle table
    if (ws->tw30_n != n_total || ws->tw30_curve != curve) {
        if ((rc = ws->tw30.ensure(n_total * ICC30_SLOT_WORDS * 4))) return rc;
        hipLaunchKernelGGL((k_icc_twiddles30<Q>), dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, stream,
                           (const IccElem<Q>*)ws->tw.p, (uint32_t)n_total, (uint32_t*)ws->tw30.p);
        ws->tw30_n = (uint32_t)n_total;
        ws->tw30_curve = curve;
    }
    {
        ProfScope ps("icc_mix", stream);
        hipLaunchKernelGGL((k_icc_mix30<Q>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_a0, d_a1, (uint32_t)len,
                           (uint32_t)ncols, (const uint32_t*)ws->tw30.p, (uint32_t)(n_total / len), d_out);
    }
    PORLA_HIP(hipGetLastError());
    return ws->fence.leave(stream);
}

// wt = w^reverse_bits(write_step % N, height - 1) (Server.hpp:1391, 1494; Client.hpp:998, 1043) as the residue pair the data side
// multiplies by, and -- plain_be -- as the 32-byte big-endian integer the MAC side uses as a scalar (convert_ZZ_to_scalar)
template <class Q>
static IccElem<Q> icc_wt(size_t n, unsigned long long write_step, uint8_t* plain_be) {
    const int logn = ilog2u(n);
    const int height = logn + 1;
    Fe<IccFp> w = icc_root(n);
    uint64_t ex = rev_bits(write_step % n, height - 1);
    uint32_t e[8] = {(uint32_t)ex, (uint32_t)(ex >> 32), 0, 0, 0, 0, 0, 0};
    IccElem<Q> wt;
    wt.p = h_fe_pow<IccFp>(w, e);
    Fe<IccFp> plain = fe_from_mont<IccFp>(wt.p);
    if (plain_be)
        for (int k = 0; k < 8; k++)
            for (int b = 0; b < 4; b++) plain_be[31 - (4 * k + b)] = (uint8_t)(plain.v[k] >> (8 * b));
    Fe<Q> tq;
    for (int k = 0; k < 8; k++) tq.v[k] = plain.v[k];
    fe_reduce_plain<Q>(tq.v, 8);
    wt.q = fe_to_mont<Q>(tq);
    return wt;
}

// data side of Server::HAdd on `total` chunks (Server.hpp:1396-1398 data_B2[i] *= wt, then align_MAC's scalar part, :531-541):
// element-wise, no butterfly -- the load and finish steps of the encode with nothing in between
template <class Q>
static int icc_scale_align_core(IccWs* ws, const uint8_t* d_in, size_t total, size_t n_total, unsigned long long write_step,
                                IccOut out, uint8_t* wt_plain_be, hipStream_t stream) {
    int rc;
    if ((rc = ws->work.ensure(total * sizeof(IccElem<Q>)))) return rc;
    IccElem<Q> wt = icc_wt<Q>(n_total, write_step, wt_plain_be);
    hipLaunchKernelGGL((k_icc_load<Q>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_in, (IccElem<Q>*)ws->work.p, total, wt, 1);
    hipLaunchKernelGGL((k_icc_finish<Q>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const IccElem<Q>*)ws->work.p, total, out);
    PORLA_HIP(hipGetLastError());
    return PORLA_OK;
}

// out_y != nullptr (part must be 0): the Y part's outputs from the SAME network -- Y_k = wt X_k mod LCM (icc30_split.hip.h, XY)
template <class Q>
static int icc_encode_core(IccWs* ws, int curve, const uint8_t* d_rows, size_t n, size_t ncols, unsigned long long write_step,
                           int part, uint8_t* d_x, uint8_t* d_al, uint8_t* d_sc, int scalar_le, hipStream_t stream,
                           uint8_t* d_qres = nullptr, const IccOut* out_y = nullptr) {
    const int logn = ilog2u(n);
    if (n < 2 || ((size_t)1 << logn) != n || n > (1u << 30) || ncols == 0) {
        set_last_error("porla: ICC encode needs a power-of-two row count >= 2");
        return PORLA_ERR_ARG;
    }
    const size_t total = n * ncols;
    int rc;
    if ((rc = ws->work.ensure(total * sizeof(IccElem<Q>)))) return rc;
    if ((rc = ensure_twiddles<Q>(ws, curve, n, stream))) return rc;
    // ---- init scaling: wt = w^reverse_bits(write_step % N, height-1) for the Y part (Server.hpp:1494), 1 for X
    IccElem<Q> wt;
    wt.p = fe_one<IccFp>();
    wt.q = fe_one<Q>();
    int use_wt = 0;
    if (part == 1) { wt = icc_wt<Q>(n, write_step, nullptr); use_wt = 1; }
    IccOut out{d_x, d_al, d_sc, d_qres, scalar_le};
    if (out_y) {
        if (part != 0) { set_last_error("porla: the two-part encode takes part = 0"); return PORLA_ERR_ARG; }
        wt = icc_wt<Q>(n, write_step, nullptr);        // handed to the last pass; the network itself runs unscaled (use_wt = 0)
        if (!out_y->al && (out_y->x || out_y->sc) && (rc = ws->park_y.ensure(total * 32))) return rc;
    }
    const IccOut oy = out_y ? *out_y : IccOut{nullptr, nullptr, nullptr, nullptr, 0};
    // ceil(logn / 8) passes of (almost) equal stage counts, each through LDS tiles of ICC_TILE_ELEMS = 1 024 symbols (icc30_split.hip.h: one plane at
    // a time, two stages per LDS round trip); the first pass reads the raw chunks, the last one writes the outputs: the two
    // residue planes (9 words per symbol each) only travel between passes.
    if ((rc = ws->work.ensure(total * ICC30_PACK_WORDS * 4))) return rc;
    if (ws->tw30s_n != n || ws->tw30s_curve != curve) {
        if ((rc = ws->tw30p.ensure(n * ICC30_PSLOT_WORDS * 4)) || (rc = ws->tw30q.ensure(n * ICC30_PSLOT_WORDS * 4))) return rc;
        hipLaunchKernelGGL((k_icc_twiddles30_planes<Q>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const IccElem<Q>*)ws->tw.p, (uint32_t)n, (uint32_t*)ws->tw30p.p, (uint32_t*)ws->tw30q.p);
        ws->tw30s_n = (uint32_t)n;
        ws->tw30s_curve = curve;
    }
    // at most ICC_TILE_LOG - 1 stages per pass (a tile keeps two columns of a row side by side): 2^9 rows are one pass, 2^10 .. 2^18
    // two, beyond that three
    constexpr int max_ns = ICC_TILE_LOG - 1;
    const int passes = (logn + max_ns - 1) / max_ns;
    int s = 1;
    for (int pz = 0; pz < passes; pz++) {
        const int ns = (logn - (s - 1) + (passes - pz) - 1) / (passes - pz);
        int cc_log = ICC_TILE_LOG - ns;                                    // 2^ns rows x 2^cc_log columns = ICC_TILE_ELEMS symbols
        while (cc_log > 0 && ((size_t)1 << (cc_log - 1)) >= ncols) cc_log--;   // no wider than the row
        const size_t col_tiles = (ncols + ((size_t)1 << cc_log) - 1) >> cc_log;
        const dim3 grid((unsigned)(col_tiles * (n >> ns)));
        const bool first = pz == 0, last = pz == passes - 1;
        ProfScope ps("icc_fused", stream, true);
#define PORLA_ICC_LAUNCH(F, L)                                                                                              \
    do {                                                                                                                    \
        if (L && out_y)                                                                                                     \
            hipLaunchKernelGGL((k_icc_split30<Q, F, L, L>), grid, dim3(ICC30_SPLIT_THREADS), 0, stream, (uint32_t*)ws->work.p, \
                               (uint32_t*)ws->work.p + total * ICC30_PLANE_WORDS, (const uint32_t*)ws->tw30p.p,            \
                               (const uint32_t*)ws->tw30q.p, (uint32_t)n, (uint32_t)ncols, s, ns, cc_log, d_rows, wt, use_wt, out, oy, \
                               (uint32_t*)ws->park_y.p);                                                                    \
        else                                                                                                                \
            hipLaunchKernelGGL((k_icc_split30<Q, F, L>), grid, dim3(ICC30_SPLIT_THREADS), 0, stream, (uint32_t*)ws->work.p, \
                               (uint32_t*)ws->work.p + total * ICC30_PLANE_WORDS, (const uint32_t*)ws->tw30p.p,            \
                               (const uint32_t*)ws->tw30q.p, (uint32_t)n, (uint32_t)ncols, s, ns, cc_log, d_rows, wt, use_wt, out, oy, \
                               (uint32_t*)nullptr);                                                                         \
    } while (0)
        if (first && last) PORLA_ICC_LAUNCH(true, true);
        else if (first) PORLA_ICC_LAUNCH(true, false);
        else if (last) PORLA_ICC_LAUNCH(false, true);
        else PORLA_ICC_LAUNCH(false, false);
#undef PORLA_ICC_LAUNCH
        s += ns;
    }
    PORLA_HIP(hipGetLastError());
    return PORLA_OK;
}

static int icc_encode_dispatch(IccWs* ws, int curve, const uint8_t* d_rows, size_t n, size_t ncols, unsigned long long ws_step,
                               int part, uint8_t* d_x, uint8_t* d_al, uint8_t* d_sc, int scalar_le, hipStream_t stream,
                               uint8_t* d_qres = nullptr, const IccOut* out_y = nullptr) {
    if (curve != 0 && curve != 1) {
        set_last_error("porla: curve must be 0 (BN254 / KZG) or 1 (secp256k1 / IPA)");
        return PORLA_ERR_ARG;
    }
    int rc = ws->fence.enter(stream);      // an earlier encode on another stream may still use work / the twiddles
    if (rc) return rc;
    rc = curve == 0 ? icc_encode_core<IccBn254Fr>(ws, 0, d_rows, n, ncols, ws_step, part, d_x, d_al, d_sc, scalar_le, stream, d_qres, out_y)
                    : icc_encode_core<IccSecp256k1Fn>(ws, 1, d_rows, n, ncols, ws_step, part, d_x, d_al, d_sc, scalar_le, stream, d_qres, out_y);
    if (rc) return rc;
    return ws->fence.leave(stream);
}

int icc_wt_scalar_be(size_t n_total, unsigned long long write_step, uint8_t out[32]) {
    const int ln = ilog2u(n_total);
    if (n_total < 2 || ((size_t)1 << ln) != n_total || !out) { set_last_error("porla: n_total must be a power of two >= 2"); return PORLA_ERR_ARG; }
    (void)icc_wt<IccBn254Fr>(n_total, write_step, out);
    return PORLA_OK;
}

// The butterfly network as a matrix over Z_q: row k = the coefficients F[k][0..n) with out_k = sum_i F[k][i] * in_i
// (part 1: inputs pre-scaled by wt), 32-byte big-endian each -- the data-side encode applied to the N x N identity.
int icc_network_matrix_device(int curve, size_t n, unsigned long long write_step, int part, uint8_t* d_rows_out,
                              hipStream_t stream) {
    int rc = ensure_device();
    if (rc) return rc;
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
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
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    return icc_encode_dispatch(ws, curve, (const uint8_t*)d_rows_in, n_rows, n_cols, write_step, part, (uint8_t*)d_x_out,
                               (uint8_t*)d_aligned_out, (uint8_t*)d_scalars_out, scalar_le, (hipStream_t)stream);
}

int porla_icc_encode_xy_device(const void* d_rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                               void* d_x_out, void* d_aligned_out, void* d_scalars_out, void* d_y_x_out, void* d_y_aligned_out,
                               void* d_y_scalars_out, int scalar_le, void* stream) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!d_rows_in) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    const IccOut oy{(uint8_t*)d_y_x_out, (uint8_t*)d_y_aligned_out, (uint8_t*)d_y_scalars_out, nullptr, scalar_le};
    return icc_encode_dispatch(ws, curve, (const uint8_t*)d_rows_in, n_rows, n_cols, write_step, 0, (uint8_t*)d_x_out,
                               (uint8_t*)d_aligned_out, (uint8_t*)d_scalars_out, scalar_le, (hipStream_t)stream, nullptr, &oy);
}

// columns [c0, c1) of row-major host rows: strided upload into a compact n_rows x (c1 - c0) image, encode, strided download into
// the callers' full-width outputs (only those columns are written)
static int icc_encode_cols_host(const uint8_t* rows_in, size_t n_rows, size_t n_cols, size_t c0, size_t c1, int curve,
                                unsigned long long write_step, int part, uint8_t* x_out, uint8_t* aligned_out, uint8_t* scalars_out,
                                int scalar_le) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!rows_in || c0 >= c1 || c1 > n_cols) { set_last_error("porla: bad argument (null rows or empty / out-of-range column range)"); return PORLA_ERR_ARG; }
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    const size_t nc = c1 - c0, total = n_rows * nc;
    if ((rc = ws->in.ensure(total * 32))) return rc;
    if (x_out && (rc = ws->xo.ensure(total * 64))) return rc;
    if (aligned_out && (rc = ws->al.ensure(total * 32))) return rc;
    if (scalars_out && (rc = ws->sc.ensure(total * 32))) return rc;
    hipStream_t s = engine_stream();
    if ((rc = ws->fence.enter(s))) return rc;
    PORLA_HIP(hipMemcpy2DAsync(ws->in.p, nc * 32, rows_in + c0 * 32, n_cols * 32, nc * 32, n_rows, hipMemcpyHostToDevice, s));
    rc = icc_encode_dispatch(ws, curve, (const uint8_t*)ws->in.p, n_rows, nc, write_step, part,
                             x_out ? (uint8_t*)ws->xo.p : nullptr, aligned_out ? (uint8_t*)ws->al.p : nullptr,
                             scalars_out ? (uint8_t*)ws->sc.p : nullptr, scalar_le, s);
    if (rc) return rc;
    if (x_out) PORLA_HIP(hipMemcpy2DAsync(x_out + c0 * 64, n_cols * 64, ws->xo.p, nc * 64, nc * 64, n_rows, hipMemcpyDeviceToHost, s));
    if (aligned_out) PORLA_HIP(hipMemcpy2DAsync(aligned_out + c0 * 32, n_cols * 32, ws->al.p, nc * 32, nc * 32, n_rows, hipMemcpyDeviceToHost, s));
    if (scalars_out) PORLA_HIP(hipMemcpy2DAsync(scalars_out + c0 * 32, n_cols * 32, ws->sc.p, nc * 32, nc * 32, n_rows, hipMemcpyDeviceToHost, s));
    PORLA_HIP(hipStreamSynchronize(s));
    return PORLA_OK;
}

int porla_icc_encode_host(const uint8_t* rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                          int part, uint8_t* x_out, uint8_t* aligned_out, uint8_t* scalars_out, int scalar_le) {
    return icc_encode_cols_host(rows_in, n_rows, n_cols, 0, n_cols, curve, write_step, part, x_out, aligned_out, scalars_out, scalar_le);
}

int porla_icc_encode_cols_host(const uint8_t* rows_in, size_t n_rows, size_t n_cols, size_t col_begin, size_t col_end, int curve,
                               unsigned long long write_step, int part, uint8_t* x_out, uint8_t* aligned_out, uint8_t* scalars_out,
                               int scalar_le) {
    return icc_encode_cols_host(rows_in, n_rows, n_cols, col_begin, col_end, curve, write_step, part, x_out, aligned_out, scalars_out,
                                scalar_le);
}

// the reference splits the columns of a stage over its 8 pool threads (Server.hpp:1564-1686); here device g of `devices`
// takes the column block [g C / G, (g+1) C / G) -- 16 columns each for 128 columns on 8 GPUs -- from its own
// host thread: the 128 transforms are independent, nothing is exchanged
int porla_icc_encode_host_multi(const uint8_t* rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                                int part, uint8_t* x_out, uint8_t* aligned_out, uint8_t* scalars_out, int scalar_le, int devices) {
    int rc = ensure_device();
    if (rc) return rc;
    int visible = 0, first = 0;
    PORLA_HIP(hipGetDeviceCount(&visible));
    PORLA_HIP(hipGetDevice(&first));
    int G = devices <= 0 ? visible : (devices < visible ? devices : visible);
    if ((size_t)G > n_cols) G = (int)n_cols;
    if (G < 1) G = 1;
    std::vector<int> rcs((size_t)G, PORLA_OK);
    std::vector<std::string> errs((size_t)G);
    auto worker = [&](int g) {
        if (hipSetDevice((first + g) % visible) != hipSuccess) { rcs[g] = PORLA_ERR_HIP; errs[g] = "porla: hipSetDevice failed"; return; }
        size_t col_begin, col_end;
        porla_shard_range(n_cols, g, G, &col_begin, &col_end);
        rcs[g] = icc_encode_cols_host(rows_in, n_rows, n_cols, col_begin, col_end, curve, write_step, part, x_out, aligned_out, scalars_out, scalar_le);
        if (rcs[g]) errs[g] = porla_gpu_last_error();
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; g++) th.emplace_back(worker, g);
    worker(0);
    for (auto& t : th) t.join();
    if (G > 1) (void)hipSetDevice(first);
    for (int g = 0; g < G; g++) if (rcs[g]) { set_last_error(errs[g]); return rcs[g]; }
    return PORLA_OK;
}

// Server::HAdd, data side of one incoming block (Server.hpp:1388-1398 + align_MAC :531-541 / :495-504)
int porla_icc_hadd_host(const uint8_t* data_in, size_t n_cols, size_t n_total, unsigned long long write_step, int curve,
                        uint8_t* data_b2_out, uint8_t* scalars_out, int scalar_le, uint8_t wt_scalar_out[32]) {
    int rc = ensure_device();
    if (rc) return rc;
    const int ln = ilog2u(n_total);
    if (!data_in || n_cols == 0 || n_total < 2 || ((size_t)1 << ln) != n_total || (curve != 0 && curve != 1)) {
        set_last_error("porla: bad argument to porla_icc_hadd_host (n_total = num_blocks, a power of two >= 2)");
        return PORLA_ERR_ARG;
    }
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    if ((rc = ws->in.ensure(n_cols * 32))) return rc;
    if ((rc = ws->al.ensure(n_cols * 32))) return rc;
    if ((rc = ws->sc.ensure(n_cols * 32))) return rc;
    hipStream_t s = engine_stream();
    if ((rc = ws->fence.enter(s))) return rc;
    PORLA_HIP(hipMemcpyAsync(ws->in.p, data_in, n_cols * 32, hipMemcpyHostToDevice, s));
    IccOut out{nullptr, data_b2_out ? (uint8_t*)ws->al.p : nullptr, scalars_out ? (uint8_t*)ws->sc.p : nullptr, nullptr, scalar_le};
    rc = curve == 0 ? icc_scale_align_core<IccBn254Fr>(ws, (const uint8_t*)ws->in.p, n_cols, n_total, write_step, out, wt_scalar_out, s)
                    : icc_scale_align_core<IccSecp256k1Fn>(ws, (const uint8_t*)ws->in.p, n_cols, n_total, write_step, out, wt_scalar_out, s);
    if (rc) return rc;
    if (data_b2_out) PORLA_HIP(hipMemcpyAsync(data_b2_out, ws->al.p, n_cols * 32, hipMemcpyDeviceToHost, s));
    if (scalars_out) PORLA_HIP(hipMemcpyAsync(scalars_out, ws->sc.p, n_cols * 32, hipMemcpyDeviceToHost, s));
    if ((rc = ws->fence.leave(s))) return rc;
    PORLA_HIP(hipStreamSynchronize(s));
    return PORLA_OK;
}

// Server::HRebuildX / HRebuildY, data part (Server.hpp:1329-1386): the chain of mixes that carries an incoming block up to `level`.
// levels[i] (i <= level) points at level i's rows: 2 * 2^i rows of n_cols 64-byte symbols, the first 2^i resident, the second 2^i
// incoming (levels[0][1] = the new block); step i mixes the two halves of level i into the incoming half of level i + 1, and at
// the end level `level`'s incoming half becomes its resident half -- all on the device, one call instead of `level` mix calls.
int porla_icc_hrebuild_host(uint8_t* const* levels, int level, size_t n_cols, size_t n_total, int curve) {
    int rc = ensure_device();
    if (rc) return rc;
    const int ln = ilog2u(n_total);
    if (!levels || level < 0 || level > 30 || n_cols == 0 || n_total < 2 || ((size_t)1 << ln) != n_total || ((size_t)1 << level) > n_total ||
        (curve != 0 && curve != 1)) {
        set_last_error("porla: bad argument to porla_icc_hrebuild_host");
        return PORLA_ERR_ARG;
    }
    for (int i = 0; i <= level; i++) if (!levels[i]) { set_last_error("porla: null level"); return PORLA_ERR_ARG; }
    const size_t row = n_cols * 64, top = (size_t)1 << level;
    if (level > 0) {
        void *d_a0 = nullptr, *d_cur = nullptr, *d_next = nullptr;
        PORLA_HIP(hipMalloc(&d_a0, (top / 2) * row));
        hipError_t e1 = hipMalloc(&d_cur, top * row), e2 = hipMalloc(&d_next, top * row);
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(d_a0); (void)hipFree(d_cur); (void)hipFree(d_next); return hip_fail(e1 != hipSuccess ? e1 : e2, "hipMalloc", __FILE__, __LINE__); }
        hipStream_t s = engine_stream();
        hipError_t e = hipMemcpyAsync(d_cur, levels[0] + row, row, hipMemcpyHostToDevice, s);
        rc = e == hipSuccess ? PORLA_OK : hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
        for (int i = 0; i < level && !rc; i++) {
            const size_t len = (size_t)1 << i;
            e = hipMemcpyAsync(d_a0, levels[i], len * row, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__); break; }
            if ((rc = porla_icc_mix_device(d_a0, d_cur, len, n_cols, n_total, curve, d_next, s))) break;
            e = hipMemcpyAsync(levels[i + 1] + 2 * len * row, d_next, 2 * len * row, hipMemcpyDeviceToHost, s);
            if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__); break; }
            void* t = d_cur; d_cur = d_next; d_next = t;
        }
        hipError_t es = hipStreamSynchronize(s);
        (void)hipFree(d_a0); (void)hipFree(d_cur); (void)hipFree(d_next);
        if (rc) return rc;
        if (es != hipSuccess) return hip_fail(es, "hipStreamSynchronize", __FILE__, __LINE__);
    }
    memcpy(levels[level], levels[level] + top * row, top * row);
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
    IccWs* ws;
    if ((rc = get_icc_ws(&ws))) return rc;
    std::lock_guard<std::mutex> lk(ws->mu);
    if (curve == 0) return icc_mix_core<IccBn254Fr>(ws, 0, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_cols, n_total, (uint8_t*)d_out, (hipStream_t)hip_stream);
    return icc_mix_core<IccSecp256k1Fn>(ws, 1, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_cols, n_total, (uint8_t*)d_out, (hipStream_t)hip_stream);
}

// Server::mix(is_x, level) in ONE call (porla/Server/Server.hpp:1209-1328): the data rows on hip_stream, both point arrays (MAC
// commitments, MAC alignments: porla_icc_mac_mix_pair_device) on a second stream beside them -- the point butterflies are a chain
// of ~200 dependent group operations (0.63 ms whatever the length up to 2^13 rows), the data part is short and wide.  Asynchronous:
// hip_stream continues when both are done; nothing waits on the host.
namespace {
struct MixSide { int device = -1; hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
std::mutex g_mix_side_mu;
std::vector<MixSide> g_mix_side;
}  // namespace
int porla_server_mix_device(const void* d_data_a0, const void* d_data_a1, const void* d_mac_a0, const void* d_mac_a1, const void* d_align_a0,
                            const void* d_align_a1, size_t len, size_t n_cols, size_t n_total, int curve, void* d_data_out, void* d_mac_out,
                            void* d_align_out, void* hip_stream) {
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    MixSide side;
    // held across fork -> side launches -> join: the events and the side stream are one set per device, and two host threads
    // mixing on different streams must not interleave their records and waits (everything below only ENQUEUES)
    std::lock_guard<std::mutex> lk(g_mix_side_mu);
    {
        MixSide* found = nullptr;
        for (auto& m : g_mix_side) if (m.device == dev) found = &m;
        if (!found) {
            MixSide m;
            m.device = dev;
            PORLA_HIP(hipStreamCreateWithFlags(&m.s, hipStreamNonBlocking));
            PORLA_HIP(hipEventCreateWithFlags(&m.fork, hipEventDisableTiming));
            PORLA_HIP(hipEventCreateWithFlags(&m.join, hipEventDisableTiming));
            g_mix_side.push_back(m);
            found = &g_mix_side.back();
        }
        side = *found;
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    PORLA_HIP(hipEventRecord(side.fork, stream));
    PORLA_HIP(hipStreamWaitEvent(side.s, side.fork, 0));
    if ((rc = porla_icc_mac_mix_pair_device(d_mac_a0, d_mac_a1, d_align_a0, d_align_a1, len, n_total, curve, d_mac_out, d_align_out, side.s))) return rc;
    PORLA_HIP(hipEventRecord(side.join, side.s));
    if ((rc = porla_icc_mix_device(d_data_a0, d_data_a1, len, n_cols, n_total, curve, d_data_out, stream))) return rc;
    PORLA_HIP(hipStreamWaitEvent(stream, side.join, 0));
    return PORLA_OK;
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
