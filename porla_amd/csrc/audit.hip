// Audit row combine on gfx950: B_j = sum_i coeff_i * row_i[j] over the <= 128*height challenged rows, followed by the
// scalar part of align_MAC on B.
//
// Reference: Server::audit, porla/Server/Server.hpp:790-828 (8 pool threads, NTL vec_ZZ += int * ZZ, no reduction), then
// align_MAC(B, combined_align) (Server.hpp:903 -> :531-541: mod = B_j % p_icc; c_j = (mod - B_j) % q; B_j = mod).  The
// rows are code symbols in the reference's own row formats: 64-byte little-endian values < LCM for the cached levels
// (utils.h:473-517) or 32-byte little-endian values < p_icc for the levels kept as 256-bit rows; coefficients are
// abs(int32) drawn from the AES-CTR stream (Server.hpp:617-621).
//
// MI355X: the encoded levels are resident in HBM (2^24 blocks x 8 KiB = 137 GB fits the 288 GB part), so an audit is a
// gather of <= 3 200 rows of 8 KiB: HBM-bound, ~26 MB.  One lane owns one of the 128 columns of a slice of the challenged
// rows and accumulates the EXACT integer (512-bit value x 32-bit coefficient, 576-bit accumulator; lanes of a wave read
// 64 consecutive 64-byte symbols = 4 KiB per row, fully coalesced); a second kernel adds the slice partials and reduces
// once per column mod p_icc and mod q.  Algorithmic bytes: 8 192 B per challenged row in, 64 B per column out.
#include "engine.hpp"
#include "icc.hip.h"

#include <vector>

namespace porla {

constexpr int ACC_LIMBS = 19;  // 512 + 32 + 32 bits of head-room for up to 2^32 rows

// partial[(b * ACC_LIMBS + limb) * n_cols + col]
static __global__ void __launch_bounds__(128)
k_audit_accumulate(const uint8_t* __restrict__ rows64, const uint64_t* __restrict__ idx64, const uint32_t* __restrict__ coef64,
                   uint32_t n64, const uint8_t* __restrict__ rows32, const uint64_t* __restrict__ idx32,
                   const uint32_t* __restrict__ coef32, uint32_t n32, uint32_t n_cols, uint32_t per_block,
                   uint32_t* __restrict__ partial) {
    const uint32_t col = blockIdx.y * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    const uint32_t lo = blockIdx.x * per_block;
    const uint32_t total = n64 + n32;
    uint32_t hi = lo + per_block < total ? lo + per_block : total;
    uint32_t acc[ACC_LIMBS];
#pragma unroll
    for (int k = 0; k < ACC_LIMBS; k++) acc[k] = 0;
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t v[16];
        uint32_t cf;
        if (i < n64) {
            const uint4* src = reinterpret_cast<const uint4*>(rows64 + (idx64[i] * n_cols + col) * 64);
            uint4 a = src[0], b = src[1], c = src[2], d = src[3];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
            cf = coef64[i];
        } else {
            const uint32_t k = i - n64;
            const uint4* src = reinterpret_cast<const uint4*>(rows32 + (idx32[k] * n_cols + col) * 32);
            uint4 a = src[0], b = src[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
            for (int t = 8; t < 16; t++) v[t] = 0;
            cf = coef32[k];
        }
        // acc += cf * v
        uint64_t carry = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            uint64_t x = (uint64_t)v[t] * cf + acc[t] + carry;
            acc[t] = (uint32_t)x;
            carry = x >> 32;
        }
#pragma unroll
        for (int t = 16; t < ACC_LIMBS; t++) {
            uint64_t x = (uint64_t)acc[t] + carry;
            acc[t] = (uint32_t)x;
            carry = x >> 32;
        }
    }
#pragma unroll
    for (int k = 0; k < ACC_LIMBS; k++) partial[((size_t)blockIdx.x * ACC_LIMBS + k) * n_cols + col] = acc[k];
}

template <class Q>
static __global__ void __launch_bounds__(128)
k_audit_finish(const uint32_t* __restrict__ partial, uint32_t n_blocks, uint32_t n_cols, uint8_t* __restrict__ exact_out,
               uint8_t* __restrict__ al_out, uint8_t* __restrict__ al_be_out, uint8_t* __restrict__ sc_out) {
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    uint32_t acc[ACC_LIMBS];
#pragma unroll
    for (int k = 0; k < ACC_LIMBS; k++) acc[k] = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < ACC_LIMBS; k++) {
            uint64_t x = (uint64_t)acc[k] + partial[((size_t)b * ACC_LIMBS + k) * n_cols + col] + carry;
            acc[k] = (uint32_t)x;
            carry = x >> 32;
        }
    }
    if (exact_out) {
        uint32_t* d = reinterpret_cast<uint32_t*>(exact_out + (size_t)col * 80);
#pragma unroll
        for (int k = 0; k < 20; k++) d[k] = k < ACC_LIMBS ? acc[k] : 0;
    }
    Fe<IccFp> P = fe_from_mont<IccFp>(icc_reduce_wide<IccFp, ACC_LIMBS>(acc));   // B mod p_icc, plain
    Fe<Q> bq = icc_reduce_wide<Q, ACC_LIMBS>(acc);                                 // B mod q, Montgomery
    if (al_out) st_fe<IccFp>(reinterpret_cast<uint32_t*>(al_out + 32 * (size_t)col), P);
    if (al_be_out) {
        uint4* q4 = reinterpret_cast<uint4*>(al_be_out + 32 * (size_t)col);
        q4[0] = make_uint4(__builtin_bswap32(P.v[7]), __builtin_bswap32(P.v[6]), __builtin_bswap32(P.v[5]), __builtin_bswap32(P.v[4]));
        q4[1] = make_uint4(__builtin_bswap32(P.v[3]), __builtin_bswap32(P.v[2]), __builtin_bswap32(P.v[1]), __builtin_bswap32(P.v[0]));
    }
    if (sc_out) {
        Fe<Q> pq;
#pragma unroll
        for (int k = 0; k < 8; k++) pq.v[k] = P.v[k];
        fe_reduce_plain<Q>(pq.v, Q::MAX_Q_P + 1);
        Fe<Q> c = fe_from_mont<Q>(fe_sub<Q>(fe_to_mont<Q>(pq), bq));   // (B mod p_icc - B) mod q
        uint4* q4 = reinterpret_cast<uint4*>(sc_out + 32 * (size_t)col);
        q4[0] = make_uint4(__builtin_bswap32(c.v[7]), __builtin_bswap32(c.v[6]), __builtin_bswap32(c.v[5]), __builtin_bswap32(c.v[4]));
        q4[1] = make_uint4(__builtin_bswap32(c.v[3]), __builtin_bswap32(c.v[2]), __builtin_bswap32(c.v[1]), __builtin_bswap32(c.v[0]));
    }
}

struct AuditWs { int device = -1; Buf partial; UseFence fence; };
static std::mutex g_audit_mu;
static std::vector<AuditWs*> g_audit_ws;

}  // namespace porla

using namespace porla;

extern "C" int porla_audit_combine_device(const void* d_rows64, const uint64_t* d_idx64, const uint32_t* d_coef64, size_t n64,
                                          const void* d_rows32, const uint64_t* d_idx32, const uint32_t* d_coef32, size_t n32,
                                          size_t n_cols, int curve, void* d_exact_out, void* d_aligned_out,
                                          void* d_aligned_be_out, void* d_scalars_out, void* hip_stream) {
    int rc = ensure_device();
    if (rc) return rc;
    if ((n64 && (!d_rows64 || !d_idx64 || !d_coef64)) || (n32 && (!d_rows32 || !d_idx32 || !d_coef32)) || n_cols == 0 ||
        n64 + n32 >= (1ull << 32) || (curve != 0 && curve != 1)) {
        set_last_error("porla: bad argument to porla_audit_combine_device");
        return PORLA_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_audit_mu);
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    AuditWs* ws = nullptr;
    for (auto* w : g_audit_ws) if (w->device == dev) ws = w;
    if (!ws) { ws = new AuditWs(); ws->device = dev; g_audit_ws.push_back(ws); }
    hipStream_t stream = (hipStream_t)hip_stream;
    const uint32_t total = (uint32_t)(n64 + n32);
    uint32_t per_block = (total + 255) / 256;
    if (per_block < 4) per_block = 4;
    const uint32_t n_blocks = total ? (total + per_block - 1) / per_block : 1;
    if ((rc = ws->partial.ensure((size_t)n_blocks * ACC_LIMBS * n_cols * 4))) return rc;
    if ((rc = ws->fence.enter(stream))) return rc;      // `partial` is shared with an earlier combine on another stream
    {
        ProfScope ps("audit_accumulate", stream);
        hipLaunchKernelGGL(k_audit_accumulate, dim3(n_blocks, (unsigned)((n_cols + 127) / 128)), dim3(128), 0, stream,
                           (const uint8_t*)d_rows64, d_idx64, d_coef64, (uint32_t)n64, (const uint8_t*)d_rows32, d_idx32, d_coef32,
                           (uint32_t)n32, (uint32_t)n_cols, per_block, (uint32_t*)ws->partial.p);
    }
    {
        ProfScope ps("audit_finish", stream);
        if (curve == 0)
            hipLaunchKernelGGL((k_audit_finish<IccBn254Fr>), dim3((unsigned)((n_cols + 127) / 128)), dim3(128), 0, stream,
                               (const uint32_t*)ws->partial.p, n_blocks, (uint32_t)n_cols, (uint8_t*)d_exact_out,
                               (uint8_t*)d_aligned_out, (uint8_t*)d_aligned_be_out, (uint8_t*)d_scalars_out);
        else
            hipLaunchKernelGGL((k_audit_finish<IccSecp256k1Fn>), dim3((unsigned)((n_cols + 127) / 128)), dim3(128), 0, stream,
                               (const uint32_t*)ws->partial.p, n_blocks, (uint32_t)n_cols, (uint8_t*)d_exact_out,
                               (uint8_t*)d_aligned_out, (uint8_t*)d_aligned_be_out, (uint8_t*)d_scalars_out);
    }
    PORLA_HIP(hipGetLastError());
    return ws->fence.leave(stream);
}
