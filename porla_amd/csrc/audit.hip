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
// 64 consecutive 64-byte symbols = 4 KiB per row, fully coalesced); the eight slices of a block meet in LDS; a second kernel
// adds the blocks' partials and reduces once per column mod p_icc and mod q.  Algorithmic bytes: 8 192 B per challenged row
// in (4 096 for a 256-bit row), 64 B per column out.
#include "engine.hpp"
#include "icc.hip.h"

#include <vector>

namespace porla {

constexpr int ACC_LIMBS = 19;  // 512 + 32 + 32 bits of head-room for up to 2^32 rows

// The accumulation: a block is AUD_SLICES row slices x 128 columns (one wave pair per slice; a wave reads 64 consecutive symbols of
// a row = 4 KiB); a lane adds coeff * symbol of its slice's rows into an exact 19-limb integer, two rows' loads in flight.  The
// slices' sums meet in LDS and leave as ONE carry-save partial per block: partial[(b * ACC_LIMBS + limb) * n_cols + col], 64-bit
// limb sums (no carry chain here; k_audit_finish runs it once per column).
constexpr int AUD_SLICES = 8;         // large challenges; AUD_SLICES_SMALL for the audit's own size: 39 KB of LDS per block, so that
constexpr int AUD_SLICES_SMALL = 4;   // its blocks fit beside the single-launch MSM pair's (105 KB per CU) instead of waiting for them
constexpr int AUD_COLS = 128;

// acc += cf * v, v = WORDS little-endian 32-bit words
template <int WORDS>
__device__ __forceinline__ void audit_mac(uint32_t (&acc)[ACC_LIMBS], const uint32_t (&v)[WORDS], uint32_t cf) {
    uint64_t carry = 0;
#pragma unroll
    for (int t = 0; t < WORDS; t++) {
        uint64_t x = (uint64_t)v[t] * cf + acc[t] + carry;
        acc[t] = (uint32_t)x;
        carry = x >> 32;
    }
#pragma unroll
    for (int t = WORDS; t < ACC_LIMBS; t++) {
        uint64_t x = (uint64_t)acc[t] + carry;
        acc[t] = (uint32_t)x;
        carry = x >> 32;
    }
}
template <int WORDS>
__device__ __forceinline__ void audit_load(uint32_t (&v)[WORDS], const uint8_t* rows, uint64_t row, uint32_t n_cols, uint32_t col) {
    const uint4* src = reinterpret_cast<const uint4*>(rows + (row * n_cols + col) * (4 * WORDS));
#pragma unroll
    for (int t = 0; t < WORDS / 4; t++) {
        const uint4 q = src[t];
        v[4 * t] = q.x; v[4 * t + 1] = q.y; v[4 * t + 2] = q.z; v[4 * t + 3] = q.w;
    }
}
// rows [a, b) of one row format (WORDS = 16: 64-byte symbols, 8: 32-byte symbols), AUD_UNROLL rows' loads in flight
#ifndef AUD_UNROLL
#define AUD_UNROLL 2   // (4 rows in flight: the same 6.1 TB/s on an 8 GiB store -- not latency-limited -- at 122 registers instead of 88)
#endif
template <int WORDS>
__device__ __forceinline__ void audit_span(uint32_t (&acc)[ACC_LIMBS], const uint8_t* __restrict__ rows, const uint64_t* __restrict__ idx,
                                           const uint32_t* __restrict__ coef, uint32_t a, uint32_t b, uint32_t n_cols, uint32_t col) {
    uint32_t i = a;
    for (; i + AUD_UNROLL <= b; i += AUD_UNROLL) {
        uint32_t v[AUD_UNROLL][WORDS];
        uint64_t r[AUD_UNROLL];
        uint32_t c[AUD_UNROLL];
#pragma unroll
        for (int u = 0; u < AUD_UNROLL; u++) { r[u] = idx[i + u]; c[u] = coef[i + u]; }
#pragma unroll
        for (int u = 0; u < AUD_UNROLL; u++) audit_load<WORDS>(v[u], rows, r[u], n_cols, col);
#pragma unroll
        for (int u = 0; u < AUD_UNROLL; u++) audit_mac<WORDS>(acc, v[u], c[u]);
    }
    for (; i < b; i++) {
        uint32_t v0[WORDS];
        audit_load<WORDS>(v0, rows, idx[i], n_cols, col);
        audit_mac<WORDS>(acc, v0, coef[i]);
    }
}

template <int SLICES>
static __global__ void __launch_bounds__(SLICES * AUD_COLS) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_audit_accumulate(const uint8_t* __restrict__ rows64, const uint64_t* __restrict__ idx64, const uint32_t* __restrict__ coef64,
                   uint32_t n64, const uint8_t* __restrict__ rows32, const uint64_t* __restrict__ idx32,
                   const uint32_t* __restrict__ coef32, uint32_t n32, uint32_t n_cols, uint32_t per_slice,
                   unsigned long long* __restrict__ partial) {
    __shared__ uint32_t sums[SLICES][ACC_LIMBS][AUD_COLS];
    const uint32_t lane_col = threadIdx.x % AUD_COLS;
    const uint32_t slice = __builtin_amdgcn_readfirstlane(threadIdx.x / AUD_COLS);
    const uint32_t col0 = blockIdx.y * AUD_COLS;
    const bool live = col0 + lane_col < n_cols;
    const uint32_t col = live ? col0 + lane_col : n_cols - 1;       // idle lanes re-read the last column and drop the result
    const uint32_t total = n64 + n32;
    const uint64_t first = ((uint64_t)blockIdx.x * SLICES + slice) * per_slice;
    const uint32_t lo = first < total ? (uint32_t)first : total;
    const uint32_t hi = first + per_slice < total ? (uint32_t)(first + per_slice) : total;
    uint32_t acc[ACC_LIMBS];
#pragma unroll
    for (int k = 0; k < ACC_LIMBS; k++) acc[k] = 0;
    // the slice's rows: first those among the 64-byte-symbol rows [0, n64), then those among the 32-byte-symbol rows
    if (lo < n64) audit_span<16>(acc, rows64, idx64, coef64, lo, hi < n64 ? hi : n64, n_cols, col);
    if (hi > n64) audit_span<8>(acc, rows32, idx32, coef32, (lo > n64 ? lo : n64) - n64, hi - n64, n_cols, col);
#pragma unroll
    for (int k = 0; k < ACC_LIMBS; k++) sums[slice][k][lane_col] = acc[k];
    __syncthreads();
    for (uint32_t item = threadIdx.x; item < ACC_LIMBS * AUD_COLS; item += SLICES * AUD_COLS) {
        const uint32_t k = item / AUD_COLS, c = item % AUD_COLS;
        unsigned long long t = 0;
#pragma unroll
        for (int sl = 0; sl < SLICES; sl++) t += sums[sl][k][c];
        if (col0 + c < n_cols) partial[((size_t)blockIdx.x * ACC_LIMBS + k) * n_cols + col0 + c] = t;
    }
}

// One block per AUD_FIN_COLS columns, a lane per (quarter of the partials, limb, column): the blocks' carry-save limb sums are added
// up (64-bit: < 2^35 per partial; eight loads in flight per lane -- the loop is L2 latency, not bandwidth), then one lane per column
// runs the carry chain and reduces the exact integer once mod p_icc and once mod q.
constexpr int AUD_FIN_COLS = 8;
constexpr int AUD_FIN_SPLIT = 4;
template <class Q>
static __global__ void __launch_bounds__(AUD_FIN_SPLIT * ACC_LIMBS * AUD_FIN_COLS)
k_audit_finish(const unsigned long long* __restrict__ partial, uint32_t n_blocks, uint32_t n_cols, uint8_t* __restrict__ exact_out,
               uint8_t* __restrict__ al_out, uint8_t* __restrict__ al_be_out, uint8_t* __restrict__ sc_out) {
    __shared__ unsigned long long limb_sum[AUD_FIN_SPLIT][ACC_LIMBS][AUD_FIN_COLS];
    {
        const uint32_t c = threadIdx.x % AUD_FIN_COLS, k = (threadIdx.x / AUD_FIN_COLS) % ACC_LIMBS;
        const uint32_t part = threadIdx.x / (AUD_FIN_COLS * ACC_LIMBS);
        const uint32_t gc = blockIdx.x * AUD_FIN_COLS + c;
        unsigned long long t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = 0;
        if (gc < n_cols) {
            const unsigned long long* src = partial + (size_t)k * n_cols + gc;
            const size_t stride = (size_t)ACC_LIMBS * n_cols;
            const uint32_t b_lo = (uint32_t)((uint64_t)n_blocks * part / AUD_FIN_SPLIT), b_hi = (uint32_t)((uint64_t)n_blocks * (part + 1) / AUD_FIN_SPLIT);
            uint32_t b = b_lo;
            for (; b + 8 <= b_hi; b += 8) {
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] += src[(size_t)(b + u) * stride];
            }
            for (; b < b_hi; b++) t[0] += src[(size_t)b * stride];
        }
        limb_sum[part][k][c] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    __syncthreads();
    const uint32_t col = blockIdx.x * AUD_FIN_COLS + threadIdx.x;
    if (threadIdx.x >= AUD_FIN_COLS || col >= n_cols) return;
    uint32_t acc[ACC_LIMBS];
    {
        unsigned long long carry = 0;
#pragma unroll
        for (int k = 0; k < ACC_LIMBS; k++) {
            // limb sums < 2^32 * (<= 8 slices * n_blocks) and the carry < 2^32 likewise: no overflow below 2^29 blocks
            unsigned long long x = carry;
#pragma unroll
            for (int part = 0; part < AUD_FIN_SPLIT; part++) x += limb_sum[part][k][threadIdx.x];
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
    // rows per slice: enough blocks for two per compute unit on a large challenge, never fewer than 4 rows per slice (the audit's
    // 3 200 rows: 100 blocks of 32 rows)
    const bool small = total <= 16384;
    const uint32_t slices = small ? AUD_SLICES_SMALL : AUD_SLICES;
    uint32_t per_slice = (total + slices * 512 - 1) / (slices * 512);
    if (per_slice < 4) per_slice = 4;
    const uint32_t per_block = per_slice * slices;
    const uint32_t n_blocks = total ? (total + per_block - 1) / per_block : 1;
    if ((rc = ws->partial.ensure((size_t)n_blocks * ACC_LIMBS * n_cols * 8))) return rc;
    if ((rc = ws->fence.enter(stream))) return rc;      // `partial` is shared with an earlier combine on another stream
    {
        ProfScope ps("audit_accumulate", stream);
        const dim3 grid(n_blocks, (unsigned)((n_cols + AUD_COLS - 1) / AUD_COLS));
        if (small)
            hipLaunchKernelGGL((k_audit_accumulate<AUD_SLICES_SMALL>), grid, dim3(AUD_SLICES_SMALL * AUD_COLS), 0, stream,
                               (const uint8_t*)d_rows64, d_idx64, d_coef64, (uint32_t)n64, (const uint8_t*)d_rows32, d_idx32, d_coef32,
                               (uint32_t)n32, (uint32_t)n_cols, per_slice, (unsigned long long*)ws->partial.p);
        else
            hipLaunchKernelGGL((k_audit_accumulate<AUD_SLICES>), grid, dim3(AUD_SLICES * AUD_COLS), 0, stream,
                               (const uint8_t*)d_rows64, d_idx64, d_coef64, (uint32_t)n64, (const uint8_t*)d_rows32, d_idx32, d_coef32,
                               (uint32_t)n32, (uint32_t)n_cols, per_slice, (unsigned long long*)ws->partial.p);
    }
    {
        ProfScope ps("audit_finish", stream);
        const dim3 fgrid((unsigned)((n_cols + AUD_FIN_COLS - 1) / AUD_FIN_COLS)), fblock(AUD_FIN_SPLIT * ACC_LIMBS * AUD_FIN_COLS);
        if (curve == 0)
            hipLaunchKernelGGL((k_audit_finish<IccBn254Fr>), fgrid, fblock, 0, stream, (const unsigned long long*)ws->partial.p, n_blocks,
                               (uint32_t)n_cols, (uint8_t*)d_exact_out, (uint8_t*)d_aligned_out, (uint8_t*)d_aligned_be_out,
                               (uint8_t*)d_scalars_out);
        else
            hipLaunchKernelGGL((k_audit_finish<IccSecp256k1Fn>), fgrid, fblock, 0, stream, (const unsigned long long*)ws->partial.p, n_blocks,
                               (uint32_t)n_cols, (uint8_t*)d_exact_out, (uint8_t*)d_aligned_out, (uint8_t*)d_aligned_be_out,
                               (uint8_t*)d_scalars_out);
    }
    PORLA_HIP(hipGetLastError());
    return ws->fence.leave(stream);
}
