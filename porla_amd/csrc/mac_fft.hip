// MAC-side ICC encode: launch logic + C ABI (include/porla_gpu.h: porla_icc_mac_encode_device / _host).
//
// Two forms of the same linear map over the group (bit-exact on the affine result):
//   ladder (the default at every N since round 5): stage by stage, four or eight lanes per butterfly up to 2^16 rows, one above
//          (mac_fft.hip.h); a stage costs one 256-bit scalar multiplication of LATENCY (0.27 ms on eight lanes);
//   matrix (N <= porla_icc_mac_set_matrix_max rows; kept as an independent formulation the tests run beside the ladder -- with the
//          round-5 ladders it is slower at every size: 16 rows 2.99 against 1.03 ms, 256 rows 3.19 / 2.20, 512 rows 3.80 / 2.51,
//          tools/bench_mac_forms.py): out_k = sum_i F[k][i] * MAC_i with F = the butterfly network as an N x N matrix over Z_q (the
//          data-side encode applied to the identity, cached per (N, curve, part, wt)); evaluated as N commitments
//          against the per-call base {MAC_i} with the batched fixed-base kernels (fixed_base.hip.h, 8-bit windows):
//          N^2 * 32 independent mixed additions instead of log2(N) dependent ladders -- throughput- not latency-bound.
// Patch site in the reference (no function boundary exists there): the MAC halves of Server::CRebuild_Cached,
// porla/Server/Server.hpp:1523-1536, 1590-1609, 1658-1676 and the Y-part twins; see INTEGRATION.md.
#include "engine.hpp"
#include <cstring>
#include "mac_fft.hip.h"
#include "icc_host.hpp"

#include <cstdlib>
#include <vector>

namespace porla {

struct MacWs {
    int device = -1;
    Buf work, work_y, tws, codes, wpow, in, out, out_y;      // codes: the twiddles' digit codes (mac_fft.hip.h:k_mac_wnaf_codes)
    uint32_t tw_n = 0, codes_n = 0;
    int tw_curve = -1, codes_curve = -1;
    // matrix form
    Buf F, mont;
    size_t F_n = 0;
    int F_curve = -1, F_part = -1;
    unsigned long long F_wt_exp = 0;
    FixedBase<Bn254G1> fb_bn;
    FixedBase<Secp256k1G> fb_secp;
    UseFence fence;   // work / twiddle / matrix buffers shared between calls that may come on different streams
};
template <class C> struct FbOf;
template <> struct FbOf<Bn254G1> { static FixedBase<Bn254G1>& get(MacWs* w) { return w->fb_bn; } };
template <> struct FbOf<Secp256k1G> { static FixedBase<Secp256k1G>& get(MacWs* w) { return w->fb_secp; } };
static std::mutex g_mac_mu;
static size_t g_matrix_max = 0;       // rows up to which the matrix form runs: none by default (round 5); porla_icc_mac_set_matrix_max changes it
// Up to 2^16 rows a stage is latency bound (one or two waves per SIMD even with four lanes per butterfly): the quad-lane kernels
// (2^16 rows: 16.7 ms against 18.5 ms with one lane per butterfly; 2^17 rows: 35.4 against 22.2); the element-wise kernels switch
// earlier (`own`).  PORLA_MAC_QUAD_MAX (log2 of the row count, default 16; 0 = one lane per butterfly everywhere) moves the
// boundary -- the tests use it to run the large-N kernels at sizes the oracle finishes in seconds.
static int macq_max_log(int own) {
    static const int v = getenv("PORLA_MAC_QUAD_MAX") ? atoi(getenv("PORLA_MAC_QUAD_MAX")) : 16;
    return v < own ? v : own;
}
static std::vector<MacWs*> g_mac_ws;

static int get_mac_ws(MacWs** out) {
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    for (auto* w : g_mac_ws) if (w->device == dev) { *out = w; return PORLA_OK; }
    MacWs* w = new MacWs();
    w->device = dev;
    g_mac_ws.push_back(w);
    *out = w;
    return PORLA_OK;
}

// plain scalars (w^e mod p_icc) mod q for e < N, resident across calls with the same N and curve
template <class Q>
static int ensure_mac_twiddles(MacWs* ws, int curve, size_t n, hipStream_t stream) {
    if (ws->tw_n == n && ws->tw_curve == curve) return PORLA_OK;
    const int logn = ilog2u(n);
    int rc;
    if ((rc = ws->tws.ensure(n * 32))) return rc;
    if ((rc = ws->wpow.ensure(64 * sizeof(Fe<IccFp>)))) return rc;
    std::vector<Fe<IccFp>> wp(logn ? logn : 1);
    Fe<IccFp> cur = icc_root(n);
    for (int i = 0; i < logn; i++) { wp[i] = cur; cur = fe_sqr<IccFp>(cur); }
    PORLA_HIP(hipMemcpyAsync(ws->wpow.p, wp.data(), logn * sizeof(Fe<IccFp>), hipMemcpyHostToDevice, stream));
    PORLA_HIP(hipStreamSynchronize(stream));  // wp is a host temporary
    ProfScope ps("mac_twiddles", stream);
    hipLaunchKernelGGL((k_mac_twiddles<Q>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (uint32_t*)ws->tws.p,
                       (uint32_t)n, (const Fe<IccFp>*)ws->wpow.p, logn);
    ws->tw_n = (uint32_t)n;
    ws->tw_curve = curve;
    ws->codes_n = 0;
    return PORLA_OK;
}

// the digit codes of the twiddles the wave-uniform stages multiply by (exponents that are multiples of 32), made once per (N, curve)
// behind the twiddle table on the same stream
template <class C>
static int ensure_mac_codes(MacWs* ws, int curve, size_t n, hipStream_t stream) {
    if (ws->codes_n == n && ws->codes_curve == curve) return PORLA_OK;
    const size_t entries = n >> MACQ_CODES_EXP_SHIFT;
    if (entries == 0) return PORLA_OK;                     // (no stage of so small a network is wave-uniform)
    int rc;
    if ((rc = ws->codes.ensure(entries * MACQ_CODES_STRIDE * sizeof(uint16_t)))) return rc;
    ProfScope ps("mac_twiddles", stream);
    hipLaunchKernelGGL((k_mac_wnaf_codes<C>), dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, stream, (const uint32_t*)ws->tws.p,
                       (uint32_t)entries, (uint16_t*)ws->codes.p);
    ws->codes_n = (uint32_t)n;
    ws->codes_curve = curve;
    return PORLA_OK;
}

// dynamic LDS of the quad-lane stage / load kernels (mac_fft.hip.h:MACQ_LDS); above 64 KiB a kernel must be told once per device
template <class C>
static size_t macq_lds_bytes() {
    using M = typename C::Fp;
    static std::mutex mu;
    static std::vector<int> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return sizeof(MacQuadLds<M>);
    std::lock_guard<std::mutex> lk(mu);
    bool seen = false;
    for (int d : done) seen = seen || d == dev;
    if (!seen) {
        const int bytes = (int)sizeof(MacQuadLds<M>);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_stage30_quad<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_stage30_quad<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_load30_quad<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_load30_quad<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_mix_quad<C>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done.push_back(dev);
    }
    return sizeof(MacQuadLds<M>);
}

template <class C>
static size_t maco_lds_bytes() {
    using M = typename C::Fp;
    static std::mutex mu;
    static std::vector<int> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return sizeof(MacOctLds<M>);
    std::lock_guard<std::mutex> lk(mu);
    bool seen = false;
    for (int d : done) seen = seen || d == dev;
    if (!seen) {
        const int bytes = (int)sizeof(MacOctLds<M>);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_stage30_oct<C>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_mix_oct<C>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mac_stage30_oct_uniform<C>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done.push_back(dev);
    }
    return sizeof(MacOctLds<M>);
}
// butterflies (or mix elements) up to which a launch leaves half the chip idle at four lanes each: eight lanes per butterfly
constexpr size_t MACO_MAX_BUTTERFLIES = 8192;      // same-box A/B against four lanes: profiles/r05_k_mac_octet_ab.txt

template <class C, class Q>
static int mac_mix_core(MacWs* ws, int curve, const uint8_t* d_a0, const uint8_t* d_a1, size_t len, size_t n_total, uint8_t* d_out,
                        hipStream_t stream, const uint8_t* d_b0 = nullptr, const uint8_t* d_b1 = nullptr, uint8_t* d_out_b = nullptr) {
    int rc;
    if ((rc = ensure_mac_twiddles<Q>(ws, curve, n_total, stream))) return rc;
    ProfScope ps("mac_mix", stream);
    const unsigned sets = d_b0 ? 2u : 1u;               // the second array pair (MAC alignments beside the MAC commitments)
    const size_t quad_max = (size_t)1 << macq_max_log(14);
    if (macq_max_log(14) > 0 && len * sets <= quad_max && len * sets <= MACO_MAX_BUTTERFLIES) {   // half the chip idle at four lanes: eight
        hipLaunchKernelGGL((k_mac_mix_oct<C>), dim3((unsigned)((len + MACO_BF - 1) / MACO_BF), sets), dim3(8 * MACO_BF), maco_lds_bytes<C>(), stream, d_a0,
                           d_a1, (uint32_t)len, (const uint32_t*)ws->tws.p, (uint32_t)(n_total / len), d_out, d_b0, d_b1, d_out_b);
        PORLA_HIP(hipGetLastError());
        return PORLA_OK;
    }
    if (macq_max_log(14) > 0 && len * sets <= quad_max) {   // latency bound: four lanes per element (k_mac_stage30_quad's ladder)
        hipLaunchKernelGGL((k_mac_mix_quad<C>), dim3((unsigned)((len + MACQ_BF - 1) / MACQ_BF), sets), dim3(4 * MACQ_BF), macq_lds_bytes<C>(), stream, d_a0,
                           d_a1, (uint32_t)len, (const uint32_t*)ws->tws.p, (uint32_t)(n_total / len), d_out, d_b0, d_b1, d_out_b);
        PORLA_HIP(hipGetLastError());
        return PORLA_OK;
    }
    if (macq_max_log(14) > 0 && sets == 2 && len <= quad_max) {
        // two arrays of a length the four-lane kernel still takes one at a time (2^14 rows: 0.63 ms each; both in one launch of
        // the one-lane kernel: 1.36 ms): two launches, one after the other
        hipLaunchKernelGGL((k_mac_mix_quad<C>), dim3((unsigned)((len + MACQ_BF - 1) / MACQ_BF), 1), dim3(4 * MACQ_BF), macq_lds_bytes<C>(), stream, d_a0, d_a1,
                           (uint32_t)len, (const uint32_t*)ws->tws.p, (uint32_t)(n_total / len), d_out, d_b0, d_b1, d_out_b);
        hipLaunchKernelGGL((k_mac_mix_quad<C>), dim3((unsigned)((len + MACQ_BF - 1) / MACQ_BF), 1), dim3(4 * MACQ_BF), macq_lds_bytes<C>(), stream, d_b0, d_b1,
                           (uint32_t)len, (const uint32_t*)ws->tws.p, (uint32_t)(n_total / len), d_out_b, d_b0, d_b1, d_out_b);
        PORLA_HIP(hipGetLastError());
        return PORLA_OK;
    }
    hipLaunchKernelGGL((k_mac_mix<C>), dim3((unsigned)((len + 63) / 64), sets), dim3(64), 0, stream, d_a0, d_a1, (uint32_t)len,
                       (const uint32_t*)ws->tws.p, (uint32_t)(n_total / len), d_out, d_b0, d_b1, d_out_b);
    PORLA_HIP(hipGetLastError());
    return PORLA_OK;
}

// d_out_y != nullptr (with part == 0): BOTH parts from one butterfly network.  The network is linear over Z_q and the Y part is
// the X part's network applied to inputs scaled by wt (Server.hpp:1494-1536: Y = wt * MAC_U, then the same stages, :1691-1830),
// so Y_k = wt * X_k as group elements: one scalar multiplication per row on the X part's work array instead of a second run of
// log2(N) dependent ladders -- the affine points, hence the 64 output bytes, are the same
template <class C, class Q>
static int mac_encode_core(MacWs* ws, int curve, const uint8_t* d_in, size_t n, unsigned long long write_step, int part,
                           uint8_t* d_out, hipStream_t stream, uint8_t* d_out_y = nullptr) {
    using M = typename C::Fp;
    const int logn = ilog2u(n);
    if (n < 2 || ((size_t)1 << logn) != n || n > (1u << 30)) {
        set_last_error("porla: MAC encode needs a power-of-two row count >= 2");
        return PORLA_ERR_ARG;
    }
    int rc;
    if (n <= g_matrix_max && d_out_y) {
        // the matrix form evaluates one cached N x N matrix per part: both parts = two runs
        if ((rc = mac_encode_core<C, Q>(ws, curve, d_in, n, write_step, 0, d_out, stream))) return rc;
        return mac_encode_core<C, Q>(ws, curve, d_in, n, write_step, 1, d_out_y, stream);
    }
    if (n <= g_matrix_max) {
        // ---- matrix form
        const unsigned long long wt_exp = part == 1 ? rev_bits(write_step % n, logn) : 0;
        if (ws->F_n != n || ws->F_curve != curve || ws->F_part != part || ws->F_wt_exp != wt_exp) {
            if ((rc = ws->F.ensure(n * n * 32))) return rc;
            ProfScope ps("mac_matrix", stream);
            if ((rc = icc_network_matrix_device(curve, n, write_step, part, (uint8_t*)ws->F.p, stream))) return rc;
            ws->F_n = n; ws->F_curve = curve; ws->F_part = part; ws->F_wt_exp = wt_exp;
        }
        if ((rc = ws->mont.ensure(n * sizeof(Affine<M>)))) return rc;
        FixedBase<C>& fb = FbOf<C>::get(ws);
        fb.keep_build_buffers = true;
        hipLaunchKernelGGL((k_points_to_mont<C, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_in,
                           (Affine<M>*)ws->mont.p, (uint32_t)n);
        if ((rc = fb.build((const Affine<M>*)ws->mont.p, n, 8, stream))) return rc;      // 8-bit windows: the table is rebuilt per call
        return fb.commit_device((const uint8_t*)ws->F.p, n, n, n * 32, d_out, stream);
    }
    if ((rc = ws->work.ensure(n * sizeof(XYZZ<M>)))) return rc;
    if ((rc = ensure_mac_twiddles<Q>(ws, curve, n, stream))) return rc;
    if ((rc = ensure_mac_codes<C>(ws, curve, n, stream))) return rc;
    const uint16_t* codes = (const uint16_t*)ws->codes.p;
    int use_wt = 0;
    MacScalar wt{};
    if (part == 1 || d_out_y) {
        // wt as the group sees it: the integer (wt mod p_icc) reduced mod the group order (Server.hpp:1494-1503); a kernel
        // argument, so the call stays asynchronous
        Fe<IccFp> plain = fe_from_mont<IccFp>(icc_wt(n, write_step));
        for (int i = 0; i < 8; i++) wt.v[i] = plain.v[i];
        fe_reduce_plain<Q>(wt.v, 8);
        use_wt = part == 1;
    }
    const bool quad_path = macq_max_log(16) > 0 && n <= ((size_t)1 << macq_max_log(16));
    {
        ProfScope ps("mac_load", stream);
        if (use_wt && macq_max_log(14) > 0 && n <= ((size_t)1 << macq_max_log(14)))
            hipLaunchKernelGGL((k_mac_load30_quad<C>), dim3((unsigned)((n + MACQ_BF - 1) / MACQ_BF)), dim3(4 * MACQ_BF), macq_lds_bytes<C>(), stream, d_in,
                               (uint32_t)n, (XYZZ<M>*)ws->work.p, wt);
        else if (use_wt)
            hipLaunchKernelGGL((k_mac_load30<C, true>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_in, (uint32_t)n,
                               (XYZZ<M>*)ws->work.p, wt);
        else
            hipLaunchKernelGGL((k_mac_load30<C, false>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_in, (uint32_t)n,
                               (XYZZ<M>*)ws->work.p, wt);
    }
    for (int s = 1; s <= logn; s++) {
        ProfScope ps("mac_stage", stream);
        if (quad_path && s == 1)      // every twiddle of stage 1 is w^0 = 1: two additions per butterfly, no ladder
            hipLaunchKernelGGL((k_mac_stage1_quad<C>), dim3((unsigned)((n / 2 + 63) / 64)), dim3(256), 0, stream, (XYZZ<M>*)ws->work.p,
                               (uint32_t)n);
        else if (quad_path && (n >> s) >= 16 && n >= 128 && n / 2 <= MACO_MAX_BUTTERFLIES)
            // ... and at most 2^13 butterflies: eight lanes each, the two half-scalar ladders in different waves
            hipLaunchKernelGGL((k_mac_stage30_oct_uniform<C>), dim3((unsigned)(n / 2 / MACO_BF)), dim3(8 * MACO_BF), maco_lds_bytes<C>(), stream,
                               (XYZZ<M>*)ws->work.p, (const uint32_t*)ws->tws.p, (uint32_t)n, s, codes);
        else if (quad_path && (n >> s) >= 16 && n >= 128)
            // >= 16 butterflies per twiddle (and whole blocks of 64): a wave's 16 quads share their scalar -- the sparse ladder
            hipLaunchKernelGGL((k_mac_stage30_quad<C, true>), dim3((unsigned)((n / 2 + MACQ_BF - 1) / MACQ_BF)), dim3(4 * MACQ_BF), macq_lds_bytes<C>(), stream,
                               (XYZZ<M>*)ws->work.p, (const uint32_t*)ws->tws.p, (uint32_t)n, s, codes);
        else if (quad_path && n / 2 <= MACO_MAX_BUTTERFLIES)
            // per-butterfly scalars and at most 2^13 butterflies: eight lanes each (the two half-scalar ladders side by side)
            hipLaunchKernelGGL((k_mac_stage30_oct<C>), dim3((unsigned)((n / 2 + MACO_BF - 1) / MACO_BF)), dim3(8 * MACO_BF), maco_lds_bytes<C>(), stream,
                               (XYZZ<M>*)ws->work.p, (const uint32_t*)ws->tws.p, (uint32_t)n, s);
        else if (quad_path)
            hipLaunchKernelGGL((k_mac_stage30_quad<C, false>), dim3((unsigned)((n / 2 + MACQ_BF - 1) / MACQ_BF)), dim3(4 * MACQ_BF), macq_lds_bytes<C>(), stream,
                               (XYZZ<M>*)ws->work.p, (const uint32_t*)ws->tws.p, (uint32_t)n, s, codes);
        else if (s > 1 && (n >> s) >= 64 && ((n / 2) & 255) == 0)
            // one lane per butterfly, >= 64 butterflies per twiddle: a wave shares its scalar -- the sparse ladder
            hipLaunchKernelGGL((k_mac_stage30<C, true>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, stream, (XYZZ<M>*)ws->work.p,
                               (const uint32_t*)ws->tws.p, (uint32_t)n, s, codes);
        else
            hipLaunchKernelGGL((k_mac_stage30<C, false>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, stream, (XYZZ<M>*)ws->work.p,
                               (const uint32_t*)ws->tws.p, (uint32_t)n, s, codes);
    }
    {
        ProfScope ps("mac_finish", stream);
        hipLaunchKernelGGL((k_mac_finish<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, (const XYZZ<M>*)ws->work.p,
                           (uint32_t)n, d_out);
    }
    if (d_out_y) {
        if ((rc = ws->work_y.ensure(n * sizeof(XYZZ<M>)))) return rc;
        {
            ProfScope ps("mac_scale", stream);
            if (macq_max_log(15) > 0 && n <= ((size_t)1 << macq_max_log(15)))
                hipLaunchKernelGGL((k_mac_load30_quad<C, true>), dim3((unsigned)((n + MACQ_BF - 1) / MACQ_BF)), dim3(4 * MACQ_BF), macq_lds_bytes<C>(), stream,
                                   (const uint8_t*)ws->work.p, (uint32_t)n, (XYZZ<M>*)ws->work_y.p, wt);
            else
                hipLaunchKernelGGL((k_mac_scale30<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, (const XYZZ<M>*)ws->work.p,
                                   (uint32_t)n, (XYZZ<M>*)ws->work_y.p, wt);
        }
        ProfScope ps("mac_finish", stream);
        hipLaunchKernelGGL((k_mac_finish<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, (const XYZZ<M>*)ws->work_y.p,
                           (uint32_t)n, d_out_y);
    }
    PORLA_HIP(hipGetLastError());
    return PORLA_OK;
}

static int mac_dispatch(MacWs* ws, int curve, const uint8_t* d_in, size_t n, unsigned long long write_step, int part,
                        uint8_t* d_out, hipStream_t stream, uint8_t* d_out_y = nullptr) {
    if (curve != 0 && curve != 1) {
        set_last_error("porla: curve must be 0 (BN254) or 1 (secp256k1)");
        return PORLA_ERR_ARG;
    }
    int rc = ws->fence.enter(stream);
    if (rc) return rc;
    rc = curve == 0 ? mac_encode_core<Bn254G1, IccBn254Fr>(ws, 0, d_in, n, write_step, part, d_out, stream, d_out_y)
                    : mac_encode_core<Secp256k1G, IccSecp256k1Fn>(ws, 1, d_in, n, write_step, part, d_out, stream, d_out_y);
    if (rc) return rc;
    return ws->fence.leave(stream);
}

}  // namespace porla

using namespace porla;

extern "C" {

int porla_icc_mac_set_matrix_max(size_t n_rows) {
    std::lock_guard<std::mutex> lk(g_mac_mu);
    g_matrix_max = n_rows;
    return PORLA_OK;
}

int porla_icc_mac_encode_device(const void* d_macs_in, size_t n_rows, int curve, unsigned long long write_step, int part,
                                void* d_macs_out, void* hip_stream) {
    if (!d_macs_in || !d_macs_out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mac_mu);
    MacWs* ws;
    if ((rc = get_mac_ws(&ws))) return rc;
    return mac_dispatch(ws, curve, (const uint8_t*)d_macs_in, n_rows, write_step, part, (uint8_t*)d_macs_out, (hipStream_t)hip_stream);
}

int porla_icc_mac_encode_xy_device(const void* d_macs_in, size_t n_rows, int curve, unsigned long long write_step, void* d_macs_x_out,
                                   void* d_macs_y_out, void* hip_stream) {
    if (!d_macs_in || !d_macs_x_out || !d_macs_y_out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mac_mu);
    MacWs* ws;
    if ((rc = get_mac_ws(&ws))) return rc;
    return mac_dispatch(ws, curve, (const uint8_t*)d_macs_in, n_rows, write_step, 0, (uint8_t*)d_macs_x_out, (hipStream_t)hip_stream,
                        (uint8_t*)d_macs_y_out);
}
int porla_icc_mac_encode_xy_host(const uint8_t* macs_in, size_t n_rows, int curve, unsigned long long write_step, uint8_t* macs_x_out,
                                 uint8_t* macs_y_out) {
    if (!macs_in || !macs_x_out || !macs_y_out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mac_mu);
    MacWs* ws;
    if ((rc = get_mac_ws(&ws))) return rc;
    if ((rc = ws->in.ensure(n_rows * 64)) || (rc = ws->out.ensure(n_rows * 64)) || (rc = ws->out_y.ensure(n_rows * 64))) return rc;
    hipStream_t s = engine_stream();
    PORLA_HIP(hipMemcpyAsync(ws->in.p, macs_in, n_rows * 64, hipMemcpyHostToDevice, s));
    rc = mac_dispatch(ws, curve, (const uint8_t*)ws->in.p, n_rows, write_step, 0, (uint8_t*)ws->out.p, s, (uint8_t*)ws->out_y.p);
    if (rc) return rc;
    PORLA_HIP(hipMemcpyAsync(macs_x_out, ws->out.p, n_rows * 64, hipMemcpyDeviceToHost, s));
    PORLA_HIP(hipMemcpyAsync(macs_y_out, ws->out_y.p, n_rows * 64, hipMemcpyDeviceToHost, s));
    PORLA_HIP(hipStreamSynchronize(s));
    return PORLA_OK;
}

int porla_icc_mac_mix_device(const void* d_a0, const void* d_a1, size_t len, size_t n_total, int curve, void* d_out, void* hip_stream) {
    int rc = ensure_device();
    if (rc) return rc;
    const int ll = ilog2u(len), ln = ilog2u(n_total);
    if (!d_a0 || !d_a1 || !d_out || len == 0 || ((size_t)1 << ll) != len || ((size_t)1 << ln) != n_total || len > n_total ||
        n_total < 2 || (curve != 0 && curve != 1)) {
        set_last_error("porla: bad argument to porla_icc_mac_mix_device (len and n_total must be powers of two, len <= n_total)");
        return PORLA_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_mac_mu);
    MacWs* ws;
    if ((rc = get_mac_ws(&ws))) return rc;
    if ((rc = ws->fence.enter((hipStream_t)hip_stream))) return rc;
    rc = curve == 0 ? mac_mix_core<Bn254G1, IccBn254Fr>(ws, 0, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_total, (uint8_t*)d_out, (hipStream_t)hip_stream)
                    : mac_mix_core<Secp256k1G, IccSecp256k1Fn>(ws, 1, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_total, (uint8_t*)d_out, (hipStream_t)hip_stream);
    if (rc) return rc;
    return ws->fence.leave((hipStream_t)hip_stream);
}

// Server::mix's two point butterflies -- MAC commitments and MAC alignments, same v^i (Server.hpp:1281-1318) -- as one launch
int porla_icc_mac_mix_pair_device(const void* d_a0, const void* d_a1, const void* d_b0, const void* d_b1, size_t len, size_t n_total,
                                  int curve, void* d_out_a, void* d_out_b, void* hip_stream) {
    int rc = ensure_device();
    if (rc) return rc;
    const int ll = ilog2u(len), ln = ilog2u(n_total);
    if (!d_a0 || !d_a1 || !d_b0 || !d_b1 || !d_out_a || !d_out_b || len == 0 || ((size_t)1 << ll) != len || ((size_t)1 << ln) != n_total ||
        len > n_total || n_total < 2 || (curve != 0 && curve != 1)) {
        set_last_error("porla: bad argument to porla_icc_mac_mix_pair_device (len and n_total must be powers of two, len <= n_total)");
        return PORLA_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_mac_mu);
    MacWs* ws;
    if ((rc = get_mac_ws(&ws))) return rc;
    if ((rc = ws->fence.enter((hipStream_t)hip_stream))) return rc;
    rc = curve == 0 ? mac_mix_core<Bn254G1, IccBn254Fr>(ws, 0, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_total, (uint8_t*)d_out_a,
                                                        (hipStream_t)hip_stream, (const uint8_t*)d_b0, (const uint8_t*)d_b1, (uint8_t*)d_out_b)
                    : mac_mix_core<Secp256k1G, IccSecp256k1Fn>(ws, 1, (const uint8_t*)d_a0, (const uint8_t*)d_a1, len, n_total, (uint8_t*)d_out_a,
                                                               (hipStream_t)hip_stream, (const uint8_t*)d_b0, (const uint8_t*)d_b1, (uint8_t*)d_out_b);
    if (rc) return rc;
    return ws->fence.leave((hipStream_t)hip_stream);
}

int porla_icc_mac_mix_host(const uint8_t* a0, const uint8_t* a1, size_t len, size_t n_total, int curve, uint8_t* out) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!a0 || !a1 || !out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    const size_t bytes = len * 64;
    void *d0 = nullptr, *d1 = nullptr, *dout = nullptr;
    PORLA_HIP(hipMalloc(&d0, bytes));
    hipError_t e1 = hipMalloc(&d1, bytes), e2 = hipMalloc(&dout, 2 * bytes);
    if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(dout); return hip_fail(e1 != hipSuccess ? e1 : e2, "hipMalloc", __FILE__, __LINE__); }
    (void)hipMemcpy(d0, a0, bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(d1, a1, bytes, hipMemcpyHostToDevice);
    rc = porla_icc_mac_mix_device(d0, d1, len, n_total, curve, dout, nullptr);
    hipError_t e3 = hipMemcpy(out, dout, 2 * bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(dout);
    if (rc) return rc;
    if (e3 != hipSuccess) return hip_fail(e3, "hipMemcpy", __FILE__, __LINE__);
    return PORLA_OK;
}

// MAC side of Server::HRebuildX / HRebuildY (MAC commitments and alignments, Server.hpp:1329-1386) and Client::HRebuildX / Y
// (Client.hpp:978-994): the same chain as porla_icc_hrebuild_host on 64-byte affine points, Client::mix (Client.hpp:921-976) per step
int porla_icc_mac_hrebuild_host(uint8_t* const* levels, int level, size_t n_total, int curve) {
    int rc = ensure_device();
    if (rc) return rc;
    const int ln = ilog2u(n_total);
    if (!levels || level < 0 || level > 30 || n_total < 2 || ((size_t)1 << ln) != n_total || ((size_t)1 << level) > n_total || (curve != 0 && curve != 1)) {
        set_last_error("porla: bad argument to porla_icc_mac_hrebuild_host");
        return PORLA_ERR_ARG;
    }
    for (int i = 0; i <= level; i++) if (!levels[i]) { set_last_error("porla: null level"); return PORLA_ERR_ARG; }
    const size_t row = 64, top = (size_t)1 << level;
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
            if ((rc = porla_icc_mac_mix_device(d_a0, d_cur, len, n_total, curve, d_next, s))) break;
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

int porla_icc_mac_encode_host(const uint8_t* macs_in, size_t n_rows, int curve, unsigned long long write_step, int part,
                              uint8_t* macs_out) {
    if (!macs_in || !macs_out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mac_mu);
    MacWs* ws;
    if ((rc = get_mac_ws(&ws))) return rc;
    if ((rc = ws->in.ensure(n_rows * 64))) return rc;
    if ((rc = ws->out.ensure(n_rows * 64))) return rc;
    hipStream_t s = engine_stream();
    PORLA_HIP(hipMemcpyAsync(ws->in.p, macs_in, n_rows * 64, hipMemcpyHostToDevice, s));
    rc = mac_dispatch(ws, curve, (const uint8_t*)ws->in.p, n_rows, write_step, part, (uint8_t*)ws->out.p, s);
    if (rc) return rc;
    PORLA_HIP(hipMemcpyAsync(macs_out, ws->out.p, n_rows * 64, hipMemcpyDeviceToHost, s));
    PORLA_HIP(hipStreamSynchronize(s));
    return PORLA_OK;
}

}  // extern "C"
