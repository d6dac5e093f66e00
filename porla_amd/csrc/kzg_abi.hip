// The 14 cgo symbols of the reference's libmultiexp.so (porla/Utils/libmultiexp.h:71-84, generated from
// porla/main.go:31-230), re-implemented over the MI355X engine.  Same names, same GoSlice ABI, same
// (absent) error behaviour: failures of the GPU path print one line and abort(), as there is no error
// channel at this boundary and silently wrong MACs would be worse.
//
//   MSMs (compute_multi_exp, compute_digest_from_srs, create_proof)  -> HIP kernels (engine.hip / msm.hip.h)
//   single-point ops, Horner evaluation, pairing check               -> host (latency-bound, 64-byte operands)
#include "engine.hpp"
#include "pairing_host.hpp"
#include "host_fold64.hpp"
#include "icc30.hip.h"
#include "../../include/libmultiexp.h"
#include "../../include/porla_gpu.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

using namespace porla;
using Fp = Bn254Fp;
using Fr = Bn254Fr;

namespace {

// compute_digest hoisted over rows (main.go:70-89): out[r] = big-endian bytes of alpha * f_r(tau) mod r, f_r given by n
// coefficients of 32 big-endian bytes (fr.SetBytes: reduced mod r).
// The evaluation with EIGHT lanes per row in the reduced-radix plain stream (rows longer than KZG_LAZY_MAX_COEFFS) of icc30.hip.h (modulus r = IccBn254Fr's q):
//   f(tau) = sum_{j<8} tau^j g_j(tau^8),   g_j(x) = sum_k c_{8k+j} x^k
// lane j runs Horner over its 16 coefficients with tau^8 in the 2^270 form -- acc * (tau^8 2^270) / 2^270 + c: the stream stays plain,
// a raw 256-bit coefficient is added unreduced (SetBytes' reduction happens in the last step), ONE product per coefficient where
// k_kzg_eval_rows spends two and a reduction -- then times tau^j, a butterfly sum over the eight lanes, and lane 0 multiplies by
// alpha and reduces once.  The 8 lanes of a row read 8 consecutive coefficients (256 B) per step, a wave 8 such runs; a row is 16
// dependent products deep instead of 128.  Bounds: acc < p + 2^248 + 2^256 < 2^258 at every step, the lane sum < 2^261.
struct KzgEvalConsts {
    uint32_t tj[8][8];   // tau^j * 2^270 mod r, canonical words, j < 8
    uint32_t t8[8];      // tau^8 * 2^270 mod r
    uint32_t alpha[8];   // alpha * 2^270 mod r
};
__global__ void __launch_bounds__(256)
k_kzg_eval_rows30(const uint8_t* __restrict__ rows, uint32_t n_rows, uint32_t n_coeffs, KzgEvalConsts K, uint8_t* __restrict__ out,
                  uint32_t out_stride, const uint8_t* __restrict__ second) {
    using Q = IccBn254Fr;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = t & 7u;
    const bool live = (t >> 3) < n_rows;
    const uint32_t r = live ? (t >> 3) : n_rows - 1;          // idle lanes redo the last row (the lane sum below is wave-wide)
    const uint8_t* row = rows + (size_t)r * n_coeffs * 32;
    const F30<Q> T8 = f30_unpack<Q>(K.t8);
    F30<Q> acc;
#pragma unroll
    for (int l = 0; l < 9; l++) acc.v[l] = 0;
    // (issuing the loads of four steps ahead of their products changes nothing: the chain is bound by its products, not its reads)
    for (uint32_t k = (n_coeffs + 7) / 8; k-- > 0;) {
        const uint32_t i = 8 * k + j;
        uint32_t c[8];
#pragma unroll
        for (int w = 0; w < 8; w++) c[w] = 0;
        if (i < n_coeffs) load_be256(c, row + (size_t)i * 32);
        acc = icc30_add<Q>(icc30_mul<Q>(acc, T8), f30_unpack<Q>(c));
    }
    uint32_t tj[8];
#pragma unroll
    for (int w = 0; w < 8; w++) {
        tj[w] = K.tj[0][w];
#pragma unroll
        for (int jj = 1; jj < 8; jj++) tj[w] = j == (uint32_t)jj ? K.tj[jj][w] : tj[w];
    }
    acc = icc30_mul<Q>(acc, f30_unpack<Q>(tj));
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        F30<Q> o;
#pragma unroll
        for (int l = 0; l < 9; l++) o.v[l] = (uint32_t)__shfl_xor((int)acc.v[l], m);
        acc = icc30_add<Q>(acc, o);
    }
    if (!live) return;
    if (j == 1 && second) {     // the MAC batch: the row's second scalar rides along (out_stride = 64)
        const uint4* src = (const uint4*)(second + (size_t)r * 32);
        uint4* dst = (uint4*)(out + (size_t)r * out_stride + 32);
        dst[0] = src[0]; dst[1] = src[1];
    }
    if (j != 0) return;
    const Fe<Q> res = icc30_canonical<Q>(icc30_reduce_top<Q>(icc30_mul<Q>(acc, f30_unpack<Q>(K.alpha))));
    store_be256(out + (size_t)r * out_stride, res.v);
}

// The evaluation as a DOT PRODUCT with the reduction left to the end: f(tau) = sum_i c_i tau^i with the powers of tau in a table
// (9 limbs of 29 bits each, staged in LDS), lane j of a row's eight taking i = 8k + j.  A coefficient times a power is 81
// multiply-adds into 17 64-bit columns and nothing else -- no reduction, no carries: limbs below 2^29 make a column's nine
// products of one coefficient < 2^61.2, so SIX coefficients accumulate before the columns are rippled (6 * 9 * 2^58 + 2^29 <
// 2^64) -- where the Horner form above pays a full modular product (81 + 90 multiply-adds and the carries) per coefficient.
// The lane's sum (< steps * 2^510) is then three 256-bit pieces hi, mid, lo joined by two products with 2^256 in the 2^270 form,
// and from there the lanes are summed, multiplied by alpha and reduced exactly as in k_kzg_eval_rows30.
constexpr int KZG_LAZY_ENTRY_WORDS = 12;      // 9 limbs + 3 words of padding: three 16-byte LDS reads per power
constexpr uint32_t KZG_LAZY_MAX_COEFFS = 1024;
constexpr uint32_t KZG_M29 = (1u << 29) - 1u;
constexpr int KZG_LAZY_GROUP = 6;              // coefficients accumulated between two ripples of the columns
// (reading all six coefficients of a group ahead of their products: 142 registers, 0.51 ms against 0.46 at 2^19 rows; one step
// ahead, as below: 0.44)
__device__ __forceinline__ void kzg_unpack29(const uint32_t w[8], uint32_t out[9]) {
#pragma unroll
    for (int l = 0; l < 9; l++) {
        const int bit = 29 * l, i = bit >> 5, sft = bit & 31;
        const uint64_t two = (uint64_t)w[i] | (i + 1 < 8 ? (uint64_t)w[i + 1] << 32 : 0ull);
        out[l] = (uint32_t)(two >> sft) & KZG_M29;
    }
}
__device__ __forceinline__ void kzg_ripple29(uint64_t (&col)[19]) {
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 19; k++) {
        const uint64_t t = col[k] + carry;
        col[k] = t & KZG_M29;
        carry = t >> 29;
    }
}
struct KzgAlpha270 { uint32_t w[8]; };       // alpha * 2^270 mod r
__global__ void __launch_bounds__(256)
k_kzg_eval_rows_lazy(const uint8_t* __restrict__ rows, uint32_t n_rows, uint32_t n_coeffs, const uint32_t* __restrict__ tau29,
                     KzgAlpha270 A, uint8_t* __restrict__ out, uint32_t out_stride, const uint8_t* __restrict__ second) {
    using Q = IccBn254Fr;
    extern __shared__ uint4 kzg_lds_tau[];            // [8 * steps][KZG_LAZY_ENTRY_WORDS] words, zero beyond n_coeffs
    const uint32_t steps = (n_coeffs + 7) / 8;
    for (uint32_t i = threadIdx.x; i < steps * 8u * (KZG_LAZY_ENTRY_WORDS / 4); i += blockDim.x)
        kzg_lds_tau[i] = reinterpret_cast<const uint4*>(tau29)[i];
    __syncthreads();
    const uint32_t j = threadIdx.x & 7u;
    const F30<Q> C526 = f30_const<Q>(Icc30Const<Q>::C526);
    // a block takes 32 rows at a time, grid-strided: the table above is staged once per block, not once per 32 rows
    for (uint32_t r0 = blockIdx.x * (blockDim.x >> 3); r0 < n_rows; r0 += gridDim.x * (blockDim.x >> 3)) {
        const uint32_t rr = r0 + (threadIdx.x >> 3);
        const bool live = rr < n_rows;
        const uint32_t r = live ? rr : n_rows - 1;            // idle lanes redo the last row (the lane sum below is wave-wide)
        const uint8_t* row = rows + (size_t)r * n_coeffs * 32;
        uint64_t col[19];
#pragma unroll
        for (int k = 0; k < 19; k++) col[k] = 0;
        uint32_t since = 0;
        uint4 nhi = make_uint4(0, 0, 0, 0), nlo = nhi;        // the next step's coefficient, read one step ahead of its products
        if (j < n_coeffs) { nhi = reinterpret_cast<const uint4*>(row + (size_t)j * 32)[0]; nlo = reinterpret_cast<const uint4*>(row + (size_t)j * 32)[1]; }
        for (uint32_t k = 0; k < steps; k++) {
            const uint32_t i = 8 * k + j;
            const uint4 chi = nhi, clo = nlo;
            nhi = make_uint4(0, 0, 0, 0); nlo = nhi;
            if (i + 8 < n_coeffs) {
                const uint4* q = reinterpret_cast<const uint4*>(row + (size_t)(i + 8) * 32);
                nhi = q[0]; nlo = q[1];
            }
            const uint32_t c[8] = {__builtin_bswap32(clo.w), __builtin_bswap32(clo.z), __builtin_bswap32(clo.y), __builtin_bswap32(clo.x),
                                   __builtin_bswap32(chi.w), __builtin_bswap32(chi.z), __builtin_bswap32(chi.y), __builtin_bswap32(chi.x)};
            uint32_t x[9];
            kzg_unpack29(c, x);
            const uint4* tp = kzg_lds_tau + (size_t)i * (KZG_LAZY_ENTRY_WORDS / 4);
            const uint4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            const uint32_t y[9] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x};
#pragma unroll
            for (int a = 0; a < 9; a++)
#pragma unroll
                for (int b = 0; b < 9; b++) col[a + b] += (uint64_t)x[a] * y[b];
            if (++since == KZG_LAZY_GROUP) { kzg_ripple29(col); since = 0; }
        }
        kzg_ripple29(col);
        // 19 limbs of 29 bits -> 17 words of 32: lo = words 0..7, mid = 8..15, hi = word 16 (the sum is below 2^517 for 1024 coefficients)
        uint32_t W[17];
#pragma unroll
        for (int w = 0; w < 17; w++) {
            const int bit = 32 * w, l = bit / 29, sft = bit % 29;
            uint64_t v = col[l] >> sft;
            if (l + 1 < 19) v |= col[l + 1] << (29 - sft);
            if (l + 2 < 19 && 58 - sft < 32) v |= col[l + 2] << (58 - sft);
            W[w] = (uint32_t)v;
        }
        uint32_t hi[8];
#pragma unroll
        for (int w = 0; w < 8; w++) hi[w] = w == 0 ? W[16] : 0u;
        F30<Q> acc = icc30_add<Q>(icc30_mul<Q>(f30_unpack<Q>(hi), C526), f30_unpack<Q>(W + 8));
        acc = icc30_add<Q>(icc30_mul<Q>(acc, C526), f30_unpack<Q>(W));
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            F30<Q> o;
#pragma unroll
            for (int l = 0; l < 9; l++) o.v[l] = (uint32_t)__shfl_xor((int)acc.v[l], m);
            acc = icc30_add<Q>(acc, o);
        }
        if (!live) continue;
        if (j == 1 && second) {     // the MAC batch: the row's second scalar rides along (out_stride = 64)
            const uint4* src = (const uint4*)(second + (size_t)r * 32);
            uint4* dst = (uint4*)(out + (size_t)r * out_stride + 32);
            dst[0] = src[0]; dst[1] = src[1];
        }
        if (j != 0) continue;
        const Fe<Q> res = icc30_canonical<Q>(icc30_reduce_top<Q>(icc30_mul<Q>(acc, f30_unpack<Q>(A.w))));
        store_be256(out + (size_t)r * out_stride, res.v);
    }
}

struct KzgState {
    std::mutex mu;
    bool have_key = false;
    Fe<Fr> tau, alpha;            // Montgomery form mod r
    uint8_t tau_raw[32] = {0};    // big.Int of the raw key bytes, reduced mod r, big-endian
    long long n_samples = 0;
    std::vector<Affine<Fp>> srs;  // SRS.G1, Montgomery form (host copy)
    unsigned long long version = 1;   // bumped whenever the SRS, the hiding base or the table window changes
    int commit_window = 0;        // 0 = automatic
    // HBM copies, one set per device that has been used (the row-range splitter of porla_kzg_commit_batch_host_multi runs one
    // host thread per device; a process pinned to one GPU only ever creates its own)
    struct Dev {
        int device = -1;
        Affine<Fp>* d_srs = nullptr;  // resident Montgomery copy of the SRS
        size_t d_srs_cap = 0;
        unsigned long long srs_version = 0, g_version = 0, h_version = 0;   // what the tables below were built from
        FixedBase<Bn254G1> fb;        // window-multiples table of the SRS (fixed_base.hip.h)
        unsigned long long gh_version = 0;
        FixedBase<Bn254G1> fb_g, fb_h;   // one-point tables of G1[0] and of the MAC hiding base (client-side batches)
        FixedBase<Bn254G1> fb_gh;        // the two of them as one 2-point table (porla_kzg_mac_batch_device)
        void* d_eval = nullptr;       // scratch: evaluated scalars of a digest batch
        size_t d_eval_cap = 0;
        void* d_tau29 = nullptr;      // powers of tau in 29-bit limbs (k_kzg_eval_rows_lazy), for the key and row length below
        Fe<Fr> tau29_tau;
        uint32_t tau29_n = 0;
    };
    Dev* devs[16] = {nullptr};
    bool have_g2 = false;
    G2Affine g2[2];               // SRS.G2[0], SRS.G2[1]
    Affine<Fp> h_mac;             // MAC hiding base (main.go:28,58-59)
};
KzgState g;

[[noreturn]] void die(const char* where, int rc) {
    fprintf(stderr, "libmultiexp (MI355X): %s failed (%d): %s\n", where, rc, porla_gpu_last_error());
    abort();
}

Affine<Fp> generator() {
    Affine<Fp> a;
    a.x = fe_zero<Fp>(); a.x.v[0] = 1; a.x = fe_to_mont<Fp>(a.x);
    a.y = fe_zero<Fp>(); a.y.v[0] = 2; a.y = fe_to_mont<Fp>(a.y);
    return a;
}

// G1Affine.Unmarshal on a 64-byte slice (main.go:130,144,198,...): flag bits 00 -> uncompressed.
// gnark would treat flag bits 10/11/01 as a compressed encoding; Porla never produces them in
// 64-byte buffers (coordinates are < p < 2^254), they are decoded the gnark way for completeness.
Affine<Fp> unmarshal64(const uint8_t* b) {
    uint8_t flags = b[0] & 0xC0;
    if (flags == 0x00) return h_affine_from_bytes<Fp>(b);
    Affine<Fp> a;
    if (flags == 0x40 || !g1_decompress(b, &a)) { a.x = fe_zero<Fp>(); a.y = fe_zero<Fp>(); }
    return a;
}

void fr_plain_be(uint8_t out[32], const Fe<Fr>& a) { h_fe_to_be<Fr>(out, a); }

// the current device's copies (g.mu held)
int current_dev(KzgState::Dev** out) {
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { set_last_error("porla: device index out of range"); return PORLA_ERR_ARG; }
    if (!g.devs[dev]) { g.devs[dev] = new KzgState::Dev(); g.devs[dev]->device = dev; }
    *out = g.devs[dev];
    return PORLA_OK;
}

// make this device's HBM copies (Montgomery SRS + its window-multiples table) current; g.mu held by the caller
int refresh_srs_locked(KzgState::Dev** out) {
    KzgState::Dev* kd;
    int rc = current_dev(&kd);
    if (rc) return rc;
    *out = kd;
    if (kd->srs_version == g.version) return PORLA_OK;
    if (g.srs.empty()) return PORLA_ERR_STATE;
    const size_t bytes = g.srs.size() * sizeof(Affine<Fp>);
    if (kd->d_srs && kd->d_srs_cap < bytes) { PORLA_HIP(hipFree(kd->d_srs)); kd->d_srs = nullptr; }
    if (!kd->d_srs) { PORLA_HIP(hipMalloc((void**)&kd->d_srs, bytes ? bytes : 64)); kd->d_srs_cap = bytes; }
    PORLA_HIP(hipMemcpy(kd->d_srs, g.srs.data(), bytes, hipMemcpyHostToDevice));
    hipStream_t s = engine_stream();
    {
        std::lock_guard<std::mutex> lk(kd->fb.mu);
        rc = kd->fb.build(kd->d_srs, g.srs.size(), g.commit_window, s);
    }
    if (rc) return rc;
    kd->srs_version = g.version;
    return PORLA_OK;
}

// kzg.Commit(f, srs) (main.go:114,164) for `n_rows` coefficient rows: fixed-base table path, len <= n_samples
int commit_rows(const uint8_t* rows, bool device_ptrs, size_t n_rows, size_t len, uint8_t* out, hipStream_t stream, bool guest_room = false) {
    // lock order: the state, then the table; the table's mutex is taken BEFORE the state is let go, so that neither
    // init_SRS_from_data / porla_kzg_set_commit_window nor porla_kzg_release_device_memory can rebuild or free the table between
    // the checks and the commit (compute_digest_from_srs comes from 8 pool threads, Server.hpp:550-560)
    std::unique_lock<std::mutex> lk(g.mu);
    KzgState::Dev* kd = nullptr;
    int rc = refresh_srs_locked(&kd);
    if (rc) {
        if (rc == PORLA_ERR_STATE) set_last_error("porla: SRS not initialised (call init_SRS / init_SRS_from_data first)");
        return rc;
    }
    if (len > g.srs.size()) { set_last_error("porla: more coefficients than SRS points"); return PORLA_ERR_STATE; }
    std::unique_lock<std::mutex> lkfb(kd->fb.mu);
    lk.unlock();
    if (device_ptrs) return kd->fb.commit_device(rows, n_rows, len, len * 32, out, stream, guest_room);
    return kd->fb.commit_host(rows, n_rows, len, len * 32, out, engine_stream());
}

// compute_digest_from_srs arrives ONE row per call from up to 8 pool threads at once (Server.hpp:550-560, 1054-1078, 1530-1535):
// calls that meet here are coalesced -- whoever finds no batch in progress becomes the leader, takes every row queued so far
// (its own included), commits them in ONE launch (FixedBase::commit_small) and hands the results back; rows that arrive while a
// batch is in flight form the next one.  A lone caller pays nothing for it (a batch of one).
struct CommitQueue {
    struct Item { const uint8_t* row; uint8_t* out; int rc; std::atomic<bool> done; std::string err; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Item*> q;
    std::atomic<bool> leader{false};
    bool contended = false;      // a caller found a batch in flight since the last batch was formed
};
CommitQueue cq;
constexpr int COMMIT_LINGER_US = 12;

int commit_coalesced(const uint8_t* row, uint8_t out[64]) {
    CommitQueue::Item it;
    it.row = row; it.out = out; it.rc = PORLA_OK; it.done.store(false);
    std::unique_lock<std::mutex> lk(cq.mu);
    cq.q.push_back(&it);
    for (;;) {
        if (it.done.load()) { if (it.rc) set_last_error(it.err); return it.rc; }
        if (cq.leader.load()) { cq.contended = true; cq.cv.wait(lk); continue; }
        cq.leader = true;
        if (cq.contended) {
            // Several threads are calling (the reference's pool threads, Server.hpp:550-560): the callers of the batch that just
            // finished are on their way back.  Without a pause the first one back leads a batch of whoever happens to be queued
            // (sizes 1 .. 8 evenly); 12 us let them form ONE batch: 59 k -> 87 k
            // commits/s from 8 C threads, 33 k -> 46 k from 4 (profiles/r02_t_commit_queue_linger.txt).  A lone caller never waits.
            cq.contended = false;
            lk.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(COMMIT_LINGER_US)) __builtin_ia32_pause();
            lk.lock();
        }
        std::vector<CommitQueue::Item*> batch;
        const size_t take = cq.q.size() < (size_t)FB_SMALL_MAX_ROWS ? cq.q.size() : (size_t)FB_SMALL_MAX_ROWS;
        batch.assign(cq.q.begin(), cq.q.begin() + (long)take);
        cq.q.erase(cq.q.begin(), cq.q.begin() + (long)take);
        lk.unlock();
        int rc;
        std::string err;
        {
            std::unique_lock<std::mutex> ls(g.mu);
            KzgState::Dev* kd = nullptr;
            rc = refresh_srs_locked(&kd);
            const size_t len = (size_t)g.n_samples;
            if (rc == PORLA_ERR_STATE) set_last_error("porla: SRS not initialised (call init_SRS / init_SRS_from_data first)");
            if (!rc && len > g.srs.size()) { set_last_error("porla: more coefficients than SRS points"); rc = PORLA_ERR_STATE; }
            if (!rc) {
                std::unique_lock<std::mutex> lf(kd->fb.mu);
                ls.unlock();
                const uint8_t* rp[FB_SMALL_MAX_ROWS];
                uint8_t* op[FB_SMALL_MAX_ROWS];
                for (size_t i = 0; i < batch.size(); i++) { rp[i] = batch[i]->row; op[i] = batch[i]->out; }
                if (FixedBase<Bn254G1>::small_ok(batch.size(), len)) {
                    rc = kd->fb.commit_small(rp, batch.size(), len, op, engine_stream());
                } else {
                    for (size_t i = 0; i < batch.size() && !rc; i++) rc = kd->fb.commit_host(rp[i], 1, len, len * 32, op[i], engine_stream());
                }
            }
            if (rc) err = porla_gpu_last_error();
        }
        lk.lock();
        for (auto* b : batch) { b->rc = rc; b->err = err; b->done.store(true, std::memory_order_release); }
        cq.leader.store(false, std::memory_order_release);
        cq.cv.notify_all();
    }
}

void copy_out(GoSlice* dst, const uint8_t* src, size_t n) {  // Go copy(): min(len(dst), len(src))
    size_t m = (size_t)(dst->len < 0 ? 0 : dst->len);
    if (m > n) m = n;
    memcpy(dst->data, src, m);
}

}  // namespace

extern "C" {

// main.go:31-40
void init_key(GoSlice* tau_key_in, GoSlice* alpha_key_in) {
    std::lock_guard<std::mutex> lk(g.mu);
    g.tau = h_fe_from_be_var<Fr>((const uint8_t*)tau_key_in->data, (size_t)tau_key_in->len);
    g.alpha = h_fe_from_be_var<Fr>((const uint8_t*)alpha_key_in->data, (size_t)alpha_key_in->len);
    fr_plain_be(g.tau_raw, g.tau);
    g.have_key = true;
}

// main.go:42-60.  kzg.NewSRS: G1[i] = tau^i * G, G2 = {G2gen, tau * G2gen}; WriteTo: 4-byte BE count,
// n compressed G1 (32 B), 2 compressed G2 (64 B) = 32n + 132 bytes (Client.hpp:350-357).
void init_SRS(GoInt SRS_size, GoSlice* out, GoInt64* out_len) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.have_key || SRS_size <= 0) { fprintf(stderr, "libmultiexp (MI355X): init_SRS before init_key\n"); abort(); }
    g.n_samples = SRS_size;
    g.srs.assign((size_t)SRS_size, Affine<Fp>());
    Affine<Fp> G = generator();
    Fe<Fr> t = fe_one<Fr>();
    std::vector<XYZZ<Fp>> proj((size_t)SRS_size);
    for (long long i = 0; i < SRS_size; i++) {
        uint32_t k[8];
        h_fe_to_plain<Fr>(k, t);
        static HostFixedBase<Fp> fb_gen;              // every power of tau multiplies the same generator: table of its multiples
        proj[(size_t)i] = fb_gen.mul(G, k);
        t = fe_mul<Fr>(t, g.tau);
    }
    h_batch_xyzz_to_affine64<Fp>(proj.data(), (size_t)SRS_size, g.srs.data());       // one inversion per 64 points
    uint32_t tau_plain[8];
    h_fe_to_plain<Fr>(tau_plain, g.tau);
    g.g2[0] = g2_generator();
    g.g2[1] = g2_scalar_mul(g.g2[0], tau_plain);
    g.have_g2 = true;

    std::vector<uint8_t> blob(4 + 32 * (size_t)SRS_size + 128);
    blob[0] = (uint8_t)(SRS_size >> 24); blob[1] = (uint8_t)(SRS_size >> 16);
    blob[2] = (uint8_t)(SRS_size >> 8);  blob[3] = (uint8_t)SRS_size;
    for (long long i = 0; i < SRS_size; i++) g1_compress(&blob[4 + 32 * (size_t)i], g.srs[(size_t)i]);
    g2_compress(&blob[4 + 32 * (size_t)SRS_size], g.g2[0]);
    g2_compress(&blob[4 + 32 * (size_t)SRS_size + 64], g.g2[1]);
    if (out_len) *out_len = (GoInt64)blob.size();
    copy_out(out, blob.data(), blob.size());

    // MAC hiding h = random * G1[0] (main.go:52-59: fr.SetRandom -> crypto/rand; non-reproducible by design)
    std::random_device rd;
    uint8_t rb[32];
    for (int i = 0; i < 32; i += 4) { uint32_t v = rd(); memcpy(rb + i, &v, 4); }
    Fe<Fr> rnd = h_fe_from_be<Fr>(rb);
    uint32_t k[8];
    h_fe_to_plain<Fr>(k, rnd);
    g.h_mac = h_xyzz_to_affine64<Fp>(h_scalar_mul64<Fp>(g.srs[0], k));

    g.version++;  // the HBM copies are rebuilt on first use by a commit (the client side never needs the GPU)
}

// main.go:62-68: SRS.ReadFrom
void init_SRS_from_data(GoInt SRS_size, GoSlice* in) {
    std::lock_guard<std::mutex> lk(g.mu);
    const uint8_t* b = (const uint8_t*)in->data;
    size_t len = (size_t)in->len;
    g.n_samples = SRS_size;
    if (len < 4) { fprintf(stderr, "libmultiexp (MI355X): init_SRS_from_data: short buffer\n"); abort(); }
    size_t cnt = ((size_t)b[0] << 24) | ((size_t)b[1] << 16) | ((size_t)b[2] << 8) | b[3];
    if (len < 4 + 32 * cnt) { fprintf(stderr, "libmultiexp (MI355X): init_SRS_from_data: short buffer\n"); abort(); }
    g.srs.assign(cnt, Affine<Fp>());
    for (size_t i = 0; i < cnt; i++) {
        if (!g1_decompress(b + 4 + 32 * i, &g.srs[i])) {
            fprintf(stderr, "libmultiexp (MI355X): init_SRS_from_data: G1[%zu] is not on the curve\n", i);
            abort();
        }
    }
    g.have_g2 = false;
    if (len >= 4 + 32 * cnt + 128) {
        g.have_g2 = g2_decompress(b + 4 + 32 * cnt, &g.g2[0]) && g2_decompress(b + 4 + 32 * cnt + 64, &g.g2[1]);
    }
    g.version++;
}

// main.go:70-89: alpha * f(tau) * G1[0] -- Horner over Fr and ONE scalar multiplication (host)
void compute_digest(GoSlice* data_in, GoSlice* data_out) {
    const uint8_t* d = (const uint8_t*)data_in->data;
    Fe<Fr> acc = fe_zero<Fr>();
    for (long long i = g.n_samples - 1; i >= 0; i--)
        acc = fe_add<Fr>(fe_mul<Fr>(acc, g.tau), h_fe_from_be<Fr>(d + 32 * i));
    acc = fe_mul<Fr>(acc, g.alpha);
    uint32_t k[8];
    h_fe_to_plain<Fr>(k, acc);
    uint8_t out[64];
    static HostFixedBase<Fp> fbG;                 // table of multiples of SRS.G1[0] (rebuilt when the SRS changes)
    const XYZZ<Fp> prod = fbG.mul(g.srs[0], k);
    h_affine_to_bytes<Fp>(out, h_xyzz_to_affine64<Fp>(prod));
    copy_out(data_out, out, 64);
}

// main.go:91-101
void compute_digest_complement(GoSlice* data_in, GoSlice* data_out) {
    Fe<Fr> s = h_fe_from_be_var<Fr>((const uint8_t*)data_in->data, (size_t)data_in->len);
    uint32_t k[8];
    h_fe_to_plain<Fr>(k, s);
    uint8_t out[64];
    static HostFixedBase<Fp> fbH;                 // table of multiples of the MAC hiding base
    const XYZZ<Fp> prod = fbH.mul(g.h_mac, k);
    h_affine_to_bytes<Fp>(out, h_xyzz_to_affine64<Fp>(prod));
    copy_out(data_out, out, 64);
}

// main.go:103-116: kzg.Commit -- GPU MSM against the resident SRS
void compute_digest_from_srs(GoSlice* data_in, GoSlice* data_out) {
    uint8_t out[64];
    int rc = commit_coalesced((const uint8_t*)data_in->data, out);
    if (rc) die("compute_digest_from_srs", rc);
    copy_out(data_out, out, 64);
}

// main.go:118-138: the large MSM -- GPU
void compute_multi_exp(GoSlice* scalars, GoSlice* points, GoInt length, GoSlice* result_out) {
    uint8_t out[64];
    int rc = porla_bn254_msm_host((const uint8_t*)scalars->data, (const uint8_t*)points->data,
                                  (size_t)(length < 0 ? 0 : length), out);
    if (rc) die("compute_multi_exp", rc);
    copy_out(result_out, out, 64);
}

// main.go:140-151
GoUint8 compare_commitment(GoSlice* commitment_a, GoSlice* commitment_b) {
    Affine<Fp> a = unmarshal64((const uint8_t*)commitment_a->data);
    Affine<Fp> b = unmarshal64((const uint8_t*)commitment_b->data);
    if (!(fe_eq<Fp>(a.x, b.x) && fe_eq<Fp>(a.y, b.y))) {
        printf("error KZG commitment\n");
        return 0;
    }
    return 1;
}

// main.go:153-175 without the commitments: y = f(z) and the quotient h = (f - y)/(X - z) of the polynomial whose n coefficients are
// given as 32-byte big-endian values (fr.SetBytes: reduced mod r); h_row receives n coefficients (the top one zero)
static void kzg_open_rows(const uint8_t* d, size_t n, unsigned long long random_point, uint8_t* h_row, uint8_t point[32], uint8_t claim[32]) {
    uint8_t zb[32] = {0};
    for (int i = 0; i < 8; i++) zb[31 - i] = (uint8_t)(random_point >> (8 * i));
    // Horner and the synthetic division in 4 x 64-bit limbs (host_fold64.hpp): the coefficients and the running values stay
    // PLAIN residues, only z is in the Montgomery form -- a Montgomery product of a plain value with z R is the plain product,
    // so the 2 n products need no conversion on either side (510 products in the 8 x 32-bit code before)
    static const Fp64<Fr> F;
    typedef Fp64<Fr>::E E64;
    auto from_be_plain = [&](const uint8_t* b) {          // fr.SetBytes: big-endian, reduced mod r
        uint64_t t[4];
        for (int i = 0; i < 4; i++) {
            uint64_t w = 0;
            for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (3 - i) + k];
            t[i] = w;
        }
        E64 v = F.cond_sub(t, 0);
        for (int q = 0; q < 5; q++) v = F.cond_sub(v.v, 0);      // 2^256 < 6 r
        return v;
    };
    auto to_be = [&](uint8_t* out, const E64& a) {
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 8; k++) out[8 * (3 - i) + k] = (uint8_t)(a.v[i] >> (8 * (7 - k)));
    };
    Fe<Fr> r2f;
    for (int i = 0; i < 8; i++) r2f.v[i] = Fr::R2[i];
    const E64 z = from_be_plain(zb);
    const E64 zM = F.mul(z, Fp64<Fr>::from(r2f));          // z R
    std::vector<E64> f(n);
    for (size_t i = 0; i < n; i++) f[i] = from_be_plain(d + 32 * i);
    E64 y;
    for (int i = 0; i < 4; i++) y.v[i] = 0;
    for (size_t i = n; i-- > 0;) y = F.add(F.mul(y, zM), f[i]);
    // synthetic division: h[n-2] = f[n-1]; h[i-1] = f[i] + z*h[i]
    memset(h_row, 0, 32 * n);
    E64 carry;
    for (int i = 0; i < 4; i++) carry.v[i] = 0;
    for (size_t i = n; i-- > 1;) {
        carry = F.add(F.mul(carry, zM), f[i]);
        to_be(&h_row[32 * (i - 1)], carry);
    }
    to_be(point, z);
    to_be(claim, y);
}

// main.go:153-175: commitment = Commit(f); y = f(z); h = (f - y)/(X - z); H = Commit(h)
void create_proof(GoUint64 random_point, GoSlice* data_in, GoSlice* commitment_out, GoSlice* proof_H,
                  GoSlice* proof_point, GoSlice* proof_claim) {
    const uint8_t* d = (const uint8_t*)data_in->data;
    size_t n = (size_t)g.n_samples;
    // Both commitments go out as ONE batch of two rows of n coefficients (h padded with a zero top coefficient: the same
    // commitment), one launch instead of two.
    std::vector<uint8_t> two(2 * 32 * n, 0);
    memcpy(two.data(), d, 32 * n);
    uint8_t zt[32], yt[32];
    kzg_open_rows(d, n, random_point, two.data() + 32 * n, zt, yt);
    uint8_t both[128];
    int rc = n ? commit_rows(two.data(), false, 2, n, both, nullptr) : PORLA_OK;
    if (n == 0) memset(both, 0, sizeof both);
    if (rc) die("create_proof", rc);
    copy_out(commitment_out, both, 64);
    copy_out(proof_H, both + 64, 64);
    copy_out(proof_point, zt, 32);
    copy_out(proof_claim, yt, 32);
}

// main.go:177-193: kzg.Verify -- e(C - y*G1, G2) == e(H, tau*G2 - z*G2), as one product of two pairings (rearranged, below)
GoUint8 verify_proof(GoSlice* commitment_in, GoSlice* proof_H, GoSlice* proof_point, GoSlice* proof_claim) {
    Affine<Fp> C = unmarshal64((const uint8_t*)commitment_in->data);
    Affine<Fp> H = unmarshal64((const uint8_t*)proof_H->data);
    Fe<Fr> z = h_fe_from_be_var<Fr>((const uint8_t*)proof_point->data, (size_t)proof_point->len);
    Fe<Fr> y = h_fe_from_be_var<Fr>((const uint8_t*)proof_claim->data, (size_t)proof_claim->len);
    if (!g.have_g2) {
        printf("Verifying is wrong\n");
        return 0;
    }
    uint32_t yk[8], zk[8];
    h_fe_to_plain<Fr>(yk, y);
    h_fe_to_plain<Fr>(zk, z);
    // e(C - y G1, G2) == e(H, (tau - z) G2)  <=>  e(C - y G1 + z H, G2) * e(-H, tau G2) == 1: the factor z moves to the G1 side,
    // where a scalar multiplication costs ~45 us (endomorphism split) instead of ~300 us in G2, and both G2 operands are the
    // SRS's own points
    XYZZ<Fp> A = h_scalar_mul64_glv<Fp, GlvBn254>(g.srs.empty() ? generator() : g.srs[0], yk);
    A.y = fe_neg<Fp>(A.y);
    if (!aff_is_inf<Fp>(C)) xyzz_madd<Fp>(A, C);
    const Affine<Fp> zH = h_xyzz_to_affine64<Fp>(h_scalar_mul64_glv<Fp, GlvBn254>(H, zk));
    if (!aff_is_inf<Fp>(zH)) xyzz_madd<Fp>(A, zH);
    Affine<Fp> Aaff = h_xyzz_to_affine64<Fp>(A);
    Affine<Fp> negH = aff_neg_if<Fp>(H, true);
    if (aff_is_inf<Fp>(H)) negH = H;
    bool ok = pairing_product_is_one(Aaff, g.g2[0], negH, g.g2[1]);
    if (!ok) {
        printf("Verifying is wrong\n");
        return 0;
    }
    return 1;
}

// main.go:195-202
void add_point(GoSlice* point_a, GoSlice* point_b) {
    Affine<Fp> a = unmarshal64((const uint8_t*)point_a->data);
    Affine<Fp> b = unmarshal64((const uint8_t*)point_b->data);
    XYZZ<Fp> p = xyzz_from_affine<Fp>(a);
    xyzz_madd<Fp>(p, b);
    uint8_t out[64];
    h_affine_to_bytes<Fp>(out, h_xyzz_to_affine64<Fp>(p));
    copy_out(point_a, out, 64);
}

// main.go:204-214
void mult_point(GoSlice* point_a, GoSlice* scalar) {
    Affine<Fp> a = unmarshal64((const uint8_t*)point_a->data);
    Fe<Fr> s = h_fe_from_be_var<Fr>((const uint8_t*)scalar->data, (size_t)scalar->len);
    uint32_t k[8];
    h_fe_to_plain<Fr>(k, s);
    uint8_t out[64];
    h_affine_to_bytes<Fp>(out, h_xyzz_to_affine64<Fp>(h_scalar_mul64_glv<Fp, GlvBn254>(a, k)));     // k is reduced (h_fe_from_be_var)
    copy_out(point_a, out, 64);
}

// main.go:216-222
void neg_point(GoSlice* point) {
    Affine<Fp> a = unmarshal64((const uint8_t*)point->data);
    if (!aff_is_inf<Fp>(a)) a.y = fe_neg<Fp>(a.y);
    uint8_t out[64];
    h_affine_to_bytes<Fp>(out, a);
    copy_out(point, out, 64);
}

// main.go:224-230
void set_inf_point(GoSlice* point) {
    uint8_t out[64] = {0};
    copy_out(point, out, 64);
}

// ---- diagnostics of the host pairing (verify_proof's engine): lets the tests check bilinearity and compare the fast
// final exponentiation / projective Miller loop with the literal reference form
static void g2_to_bytes(uint8_t out[128], const G2Affine& q) {
    if (q.inf) { memset(out, 0, 128); return; }
    h_fe_to_be<Fp>(out, q.x.a1); h_fe_to_be<Fp>(out + 32, q.x.a0); h_fe_to_be<Fp>(out + 64, q.y.a1); h_fe_to_be<Fp>(out + 96, q.y.a0);
}
static G2Affine g2_from_bytes(const uint8_t in[128]) {
    G2Affine q;
    q.x.a1 = h_fe_from_be<Fp>(in); q.x.a0 = h_fe_from_be<Fp>(in + 32); q.y.a1 = h_fe_from_be<Fp>(in + 64); q.y.a0 = h_fe_from_be<Fp>(in + 96);
    q.inf = f2_is_zero(q.x) && f2_is_zero(q.y);
    return q;
}
int porla_bn254_g2_mul_generator(const uint8_t scalar_be[32], uint8_t out[128]) {
    if (!scalar_be || !out) return PORLA_ERR_ARG;
    Fe<Fr> s = h_fe_from_be<Fr>(scalar_be);
    uint32_t k[8];
    h_fe_to_plain<Fr>(k, s);
    g2_to_bytes(out, g2_scalar_mul(g2_generator(), k));
    return PORLA_OK;
}
int porla_bn254_pairing_product_is_one(const uint8_t p1[64], const uint8_t q1[128], const uint8_t p2[64], const uint8_t q2[128],
                                       int slow) {
    if (!p1 || !q1 || !p2 || !q2) return PORLA_ERR_ARG;
    return pairing_product_is_one(h_affine_from_bytes<Fp>(p1), g2_from_bytes(q1), h_affine_from_bytes<Fp>(p2), g2_from_bytes(q2),
                                  slow != 0) ? 1 : 0;
}

// the EIP-197 predicate on its own input layout: n pairs of 192 bytes (G1 X || Y, G2 x_im || x_re || y_im || y_re, all 32-byte
// big-endian; zeros = infinity).  PORLA_ERR_ARG for what the precompile rejects (a coordinate >= p, a point off its curve, a G2
// point outside the order-r subgroup); else 1 / 0.  Pairs go two at a time through the shared Miller loop of verify_proof.
int porla_bn254_pairing_check(const uint8_t* input, size_t n_pairs, int slow) {
    if (!input && n_pairs) return PORLA_ERR_ARG;
    static const uint8_t P_BE[32] = {0x30, 0x64, 0x4e, 0x72, 0xe1, 0x31, 0xa0, 0x29, 0xb8, 0x50, 0x45, 0xb6, 0x81, 0x81, 0x58, 0x5d,
                                     0x97, 0x81, 0x6a, 0x91, 0x68, 0x71, 0xca, 0x8d, 0x3c, 0x20, 0x8c, 0x16, 0xd8, 0x7c, 0xfd, 0x47};
    static const uint32_t R_LE[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    std::vector<Affine<Fp>> ps(n_pairs);
    std::vector<G2Affine> qs(n_pairs);
    for (size_t i = 0; i < n_pairs; i++) {
        const uint8_t* in = input + 192 * i;
        for (int w = 0; w < 6; w++)
            if (memcmp(in + 32 * w, P_BE, 32) >= 0) { set_last_error("porla: pairing input coordinate >= p"); return PORLA_ERR_ARG; }
        ps[i] = h_affine_from_bytes<Fp>(in);
        if (!aff_is_inf<Fp>(ps[i])) {
            const Fe<Fp> rhs = fe_add<Fp>(fe_mul<Fp>(fe_mul<Fp>(ps[i].x, ps[i].x), ps[i].x), fp_small(3));
            if (!fe_eq<Fp>(fe_mul<Fp>(ps[i].y, ps[i].y), rhs)) { set_last_error("porla: pairing input G1 point not on the curve"); return PORLA_ERR_ARG; }
        }
        qs[i] = g2_from_bytes(in + 64);
        if (!qs[i].inf) {
            const Fp2 rhs = f2_add(f2_mul(f2_sqr(qs[i].x), qs[i].x), g2_b());
            if (!f2_eq(f2_sqr(qs[i].y), rhs)) { set_last_error("porla: pairing input G2 point not on the twist"); return PORLA_ERR_ARG; }
            if (!g2_scalar_mul(qs[i], R_LE).inf) { set_last_error("porla: pairing input G2 point not in the order-r subgroup"); return PORLA_ERR_ARG; }
        }
    }
    const Affine<Fp> p_inf = h_affine_from_bytes<Fp>(std::vector<uint8_t>(64, 0).data());
    const G2Affine q_inf{f2_zero(), f2_zero(), true};
    Fp12 f = f12_one();
    if (slow) {
        for (size_t i = 0; i < n_pairs; i++) f = f12_mul(f, miller_ate_affine(ps[i], qs[i]));
        return f12_is_one(f12_pow_final(f)) ? 1 : 0;
    }
    for (size_t i = 0; i < n_pairs; i += 2)
        f = f12_mul(f, i + 1 < n_pairs ? miller_opt_ate2(ps[i], qs[i], ps[i + 1], qs[i + 1]) : miller_opt_ate2(ps[i], qs[i], p_inf, q_inf));
    return f12_is_one(f12_final_exp(f)) ? 1 : 0;
}

// ---- client side, batched: compute_digest (main.go:70-89) and compute_digest_complement (main.go:91-101) over many rows ----
static int one_point_table(FixedBase<Bn254G1>& fb, unsigned long long& built_version, const Affine<Fp>& point) {
    if (built_version == g.version) return PORLA_OK;
    uint8_t be[64];
    h_affine_to_bytes<Fp>(be, point);
    std::lock_guard<std::mutex> lk(fb.mu);
    int rc = fb.build_from_host_bytes(be, 1, 0, engine_stream());
    if (rc) return rc;
    built_version = g.version;
    return PORLA_OK;
}

// alpha * f_r(tau) of n_rows rows into kd->d_eval at out_stride bytes per row (32, or 64 with a second scalar copied beside it).
// g.mu and the mutex of the table whose commit reads d_eval are held by the caller.
static int kzg_eval_rows_launch(KzgState::Dev* kd, const void* d_rows, size_t n_rows, uint32_t out_stride, const void* d_second,
                                hipStream_t stream) {
    int rc;
    if (kd->d_eval_cap < n_rows * out_stride) {
        if (kd->d_eval) PORLA_HIP(hipFree(kd->d_eval));      // hipFree waits for the work that still uses it
        kd->d_eval = nullptr; kd->d_eval_cap = 0;
        PORLA_HIP(hipMalloc(&kd->d_eval, n_rows * out_stride + 256));
        kd->d_eval_cap = n_rows * out_stride + 256;
    }
    // d_eval is read by the commit that follows: a previous batch on another stream must have finished with it (each table's
    // fence is recorded after its commit's last kernel; the three client-side tables share d_eval, so enter all of them)
    if ((rc = kd->fb_g.fence.enter(stream))) return rc;
    if ((rc = kd->fb_gh.fence.enter(stream))) return rc;
    // x 2^270 mod r = from_mont(x R * (2^270 R) / R)
    static constexpr uint32_t C270[8] = {0x0ffead6fu, 0x36c69455u, 0x37577218u, 0xb1e9be3cu, 0xdf11f427u, 0x9e7d8ca3u, 0xed6d3304u, 0x279be39au};   // 2^270 mod r
    const uint32_t n_coeffs = (uint32_t)g.n_samples;
    if (n_coeffs <= KZG_LAZY_MAX_COEFFS) {
        const uint32_t entries = (n_coeffs + 7) / 8 * 8;
        if (!kd->d_tau29 || kd->tau29_n != n_coeffs || memcmp(kd->tau29_tau.v, g.tau.v, sizeof(g.tau.v)) != 0) {
            std::vector<uint32_t> tab((size_t)entries * KZG_LAZY_ENTRY_WORDS, 0u);
            Fe<Fr> pw = fe_one<Fr>();
            for (uint32_t i = 0; i < n_coeffs; i++) {
                const Fe<Fr> plain = fe_from_mont<Fr>(pw);
                for (int l = 0; l < 9; l++) {
                    const int bit = 29 * l, w = bit >> 5, sft = bit & 31;
                    const uint64_t two = (uint64_t)plain.v[w] | (w + 1 < 8 ? (uint64_t)plain.v[w + 1] << 32 : 0ull);
                    tab[(size_t)i * KZG_LAZY_ENTRY_WORDS + l] = (uint32_t)(two >> sft) & KZG_M29;
                }
                pw = fe_mul<Fr>(pw, g.tau);
            }
            if (kd->d_tau29) PORLA_HIP(hipFree(kd->d_tau29));     // waits for the evaluations that still read the old table
            kd->d_tau29 = nullptr; kd->tau29_n = 0;
            PORLA_HIP(hipMalloc(&kd->d_tau29, tab.size() * 4));
            PORLA_HIP(hipMemcpy(kd->d_tau29, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            kd->tau29_tau = g.tau;
            kd->tau29_n = n_coeffs;
        }
        Fe<Fr> c270;
        for (int w = 0; w < 8; w++) c270.v[w] = C270[w];
        c270 = fe_to_mont<Fr>(c270);
        const Fe<Fr> a270 = fe_from_mont<Fr>(fe_mul<Fr>(g.alpha, c270));
        KzgAlpha270 A;
        for (int w = 0; w < 8; w++) A.w[w] = a270.v[w];
        ProfScope ps("kzg_eval_rows", stream);
        constexpr size_t lazy_grid = 2048;        // 1024 .. 16384 blocks measure the same (profiles/r03_zw_bench_digest_lazy_grid.log)
        const size_t groups = (n_rows + 31) / 32;
        hipLaunchKernelGGL(k_kzg_eval_rows_lazy, dim3((unsigned)(groups < lazy_grid ? groups : lazy_grid)), dim3(256),
                           (size_t)entries * KZG_LAZY_ENTRY_WORDS * 4, stream, (const uint8_t*)d_rows, (uint32_t)n_rows, n_coeffs,
                           (const uint32_t*)kd->d_tau29, A, (uint8_t*)kd->d_eval, out_stride, (const uint8_t*)d_second);
        return PORLA_OK;
    }
    ProfScope ps("kzg_eval_rows", stream);
    {
        // longer rows: Horner with eight lanes per row; tau^j, tau^8 and alpha in the 2^270 form
        Fe<Fr> c270;
        for (int w = 0; w < 8; w++) c270.v[w] = C270[w];
        c270 = fe_to_mont<Fr>(c270);
        auto to270 = [&](const Fe<Fr>& xm, uint32_t* dst) {
            const Fe<Fr> v = fe_from_mont<Fr>(fe_mul<Fr>(xm, c270));
            for (int w = 0; w < 8; w++) dst[w] = v.v[w];
        };
        KzgEvalConsts K;
        Fe<Fr> pw = fe_one<Fr>();
        for (int jj = 0; jj < 8; jj++) { to270(pw, K.tj[jj]); pw = fe_mul<Fr>(pw, g.tau); }
        to270(pw, K.t8);
        to270(g.alpha, K.alpha);
        hipLaunchKernelGGL(k_kzg_eval_rows30, dim3((unsigned)((8 * n_rows + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)d_rows,
                           (uint32_t)n_rows, (uint32_t)g.n_samples, K, (uint8_t*)kd->d_eval, out_stride, (const uint8_t*)d_second);
    }
    return PORLA_OK;
}

int porla_kzg_digest_batch_device(const void* d_rows, size_t n_rows, void* d_out, void* hip_stream) {
    if (n_rows && (!d_rows || !d_out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.have_key || g.srs.empty()) { set_last_error("porla: init_key / init_SRS first"); return PORLA_ERR_STATE; }
    if (n_rows == 0) return PORLA_OK;
    KzgState::Dev* kd;
    if ((rc = current_dev(&kd))) return rc;
    if ((rc = one_point_table(kd->fb_g, kd->g_version, g.srs[0]))) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    std::lock_guard<std::mutex> lk2(kd->fb_g.mu);
    if ((rc = kzg_eval_rows_launch(kd, d_rows, n_rows, 32, nullptr, stream))) return rc;
    return kd->fb_g.commit_device((const uint8_t*)kd->d_eval, n_rows, 1, 32, (uint8_t*)d_out, stream);
}

// The MAC of a block as the client forms it (Client.hpp:229/471 compute_commitment, :424-455 compute_MAC_complement, then add_point):
//   out[r] = alpha * f_r(tau) * G1[0] + s_r * h_MAC
// as ONE two-coefficient commitment per row against the table of (G1[0], h_MAC): one affine conversion per block where the two
// batches spend two and the host one more for the sum.
int porla_kzg_mac_batch_device(const void* d_rows, const void* d_scalars, size_t n_rows, void* d_out, void* hip_stream) {
    if (n_rows && (!d_rows || !d_scalars || !d_out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.have_key || g.srs.empty()) { set_last_error("porla: init_key / init_SRS first"); return PORLA_ERR_STATE; }
    if (n_rows == 0) return PORLA_OK;
    KzgState::Dev* kd;
    if ((rc = current_dev(&kd))) return rc;
    if (kd->gh_version != g.version) {
        uint8_t be[128];
        h_affine_to_bytes<Fp>(be, g.srs[0]);
        h_affine_to_bytes<Fp>(be + 64, g.h_mac);
        std::lock_guard<std::mutex> lkb(kd->fb_gh.mu);
        if ((rc = kd->fb_gh.build_from_host_bytes(be, 2, 0, engine_stream()))) return rc;
        kd->gh_version = g.version;
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    std::lock_guard<std::mutex> lk2(kd->fb_gh.mu);
    if ((rc = kzg_eval_rows_launch(kd, d_rows, n_rows, 64, d_scalars, stream))) return rc;
    return kd->fb_gh.commit_device((const uint8_t*)kd->d_eval, n_rows, 2, 64, (uint8_t*)d_out, stream);
}

int porla_kzg_complement_batch_device(const void* d_scalars, size_t n, void* d_out, void* hip_stream) {
    if (n && (!d_scalars || !d_out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.srs.empty()) { set_last_error("porla: init_SRS first (it draws the MAC hiding base)"); return PORLA_ERR_STATE; }
    if (n == 0) return PORLA_OK;
    KzgState::Dev* kd;
    if ((rc = current_dev(&kd))) return rc;
    if ((rc = one_point_table(kd->fb_h, kd->h_version, g.h_mac))) return rc;
    std::lock_guard<std::mutex> lk2(kd->fb_h.mu);
    return kd->fb_h.commit_device((const uint8_t*)d_scalars, n, 1, 32, (uint8_t*)d_out, (hipStream_t)hip_stream);
}

// the SRS size as the state holds it now (g.mu: init_SRS / init_SRS_from_data may run on another thread)
static size_t kzg_n_samples() {
    std::lock_guard<std::mutex> lk(g.mu);
    return (size_t)g.n_samples;
}

// ---- the same three batches on caller-owned host buffers: staged into a device buffer kept between calls, computed by the
// device entry on the engine's stream, copied back; blocking.  (The copies dominate: 4 KiB per block over PCIe.)
namespace {
struct ClientIo {
    std::mutex mu;                 // one host batch per device at a time: the staging buffer is shared
    void* d = nullptr;
    size_t cap = 0;
    int device = -1;
};
ClientIo g_client_io[16];

int client_batch_host(int kind /* 0 digest, 1 complement, 2 MAC */, const uint8_t* rows, const uint8_t* scalars, size_t n, uint8_t* out) {
    if (n && (!out || (kind != 1 && !rows) || (kind != 0 && !scalars))) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) return PORLA_OK;
    const size_t row_bytes = kzg_n_samples() * 32;
    if (kind != 1 && row_bytes == 0) { set_last_error("porla: init_key / init_SRS first"); return PORLA_ERR_STATE; }
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { set_last_error("porla: device index out of range"); return PORLA_ERR_STATE; }
    ClientIo& io = g_client_io[dev];
    std::lock_guard<std::mutex> lk(io.mu);
    // chunks of 16 384 blocks (64 MiB of rows): a pageable copy of that size runs at 44 GB/s, one of 512 MiB at 15
    // (tools/bench_client_host.py); one stream, so the chunks follow each other through the same staging buffer
    constexpr size_t CHUNK = 16384;
    const size_t per = n < CHUNK ? n : CHUNK;
    const size_t rows_b = kind != 1 ? per * row_bytes : 0, sc_b = kind != 0 ? per * 32 : 0, out_b = per * 64;
    const size_t half = ((rows_b + 255) & ~(size_t)255) + ((sc_b + 255) & ~(size_t)255) + ((out_b + 255) & ~(size_t)255);
    const size_t need = half;
    if (io.cap < need) {
        if (io.d) PORLA_HIP(hipFree(io.d));
        io.d = nullptr; io.cap = 0;
        PORLA_HIP(hipMalloc(&io.d, need));
        io.cap = need;
    }
    hipStream_t stream = engine_stream();
    for (size_t lo = 0; lo < n; lo += CHUNK) {
        const size_t m = n - lo < CHUNK ? n - lo : CHUNK;
        uint8_t* d_rows = (uint8_t*)io.d;
        uint8_t* d_sc = d_rows + ((rows_b + 255) & ~(size_t)255);
        uint8_t* d_out = d_sc + ((sc_b + 255) & ~(size_t)255);
        if (rows_b) PORLA_HIP(hipMemcpyAsync(d_rows, rows + lo * row_bytes, m * row_bytes, hipMemcpyHostToDevice, stream));
        if (sc_b) PORLA_HIP(hipMemcpyAsync(d_sc, scalars + lo * 32, m * 32, hipMemcpyHostToDevice, stream));
        if (kind == 0) rc = porla_kzg_digest_batch_device(d_rows, m, d_out, stream);
        else if (kind == 1) rc = porla_kzg_complement_batch_device(d_sc, m, d_out, stream);
        else rc = porla_kzg_mac_batch_device(d_rows, d_sc, m, d_out, stream);
        if (rc) { (void)hipStreamSynchronize(stream); return rc; }
        PORLA_HIP(hipMemcpyAsync(out + lo * 64, d_out, m * 64, hipMemcpyDeviceToHost, stream));
    }
    PORLA_HIP(hipStreamSynchronize(stream));
    return PORLA_OK;
}
}  // namespace
int porla_kzg_digest_batch_host(const uint8_t* rows, size_t n_rows, uint8_t* out) { return client_batch_host(0, rows, nullptr, n_rows, out); }
int porla_kzg_complement_batch_host(const uint8_t* scalars, size_t n, uint8_t* out) { return client_batch_host(1, nullptr, scalars, n, out); }
int porla_kzg_mac_batch_host(const uint8_t* rows, const uint8_t* scalars, size_t n_rows, uint8_t* out) {
    return client_batch_host(2, rows, scalars, n_rows, out);
}

// coefficients per commitment row = SRS size (0 before init_SRS*): callers that slice a row-major batch derive the row stride
// (32 bytes per coefficient) from it instead of assuming the reference's 128 (config.hpp NUM_CHUNKS)
int porla_kzg_row_coefficients(size_t* n_out) {
    if (!n_out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    *n_out = kzg_n_samples();
    return PORLA_OK;
}
// ---- batched form of compute_digest_from_srs (include/porla_gpu.h) ----
int porla_kzg_commit_batch_device(const void* d_rows, size_t n_rows, void* d_out, void* hip_stream) {
    if (n_rows && (!d_rows || !d_out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    return commit_rows((const uint8_t*)d_rows, true, n_rows, kzg_n_samples(), (uint8_t*)d_out, (hipStream_t)hip_stream);
}
// The last encode stage of a large CRebuild for the KZG build in ONE call (porla/Server/Server.hpp:1487-1833 cached, :1899-2254
// on disk): per part (X, Y): the data butterflies, align_MAC's row mod p_icc and alignment scalars (:531-541), one
// compute_digest_from_srs per row on those scalars (:550-560, :2059-2065) -- and, beside them, the MAC butterflies (:1590-1609,
// :1658-1676).  Two streams inside: the MAC network (15 dependent stages, one latency-bound wave per SIMD) starts FIRST on a side
// stream and keeps its slot on every SIMD for the length of the call, because the commitments of the 2 n rows run in the
// two-waves-per-SIMD form of their kernel (k_fb_commit<C, true>).  Asynchronous: hip_stream continues when both sides are done.
namespace {
struct StageSide { int device = -1; hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
std::mutex g_stage_side_mu;
std::vector<StageSide> g_stage_side;
}  // namespace
int porla_kzg_crebuild_stage_device(const void* d_rows_in, size_t n_rows, unsigned long long write_step, void* d_aligned_x,
                                    void* d_aligned_y, void* d_scalars_xy, void* d_commits_xy, const void* d_macs_in, void* d_macs_x,
                                    void* d_macs_y, void* hip_stream) {
    if (!d_rows_in || !d_scalars_xy || !d_commits_xy || !d_macs_in || !d_macs_x || !d_macs_y) {
        set_last_error("porla: null argument");
        return PORLA_ERR_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    const size_t n_cols = kzg_n_samples();
    if (n_cols == 0) { set_last_error("porla: SRS not initialised (call init_SRS / init_SRS_from_data first)"); return PORLA_ERR_STATE; }
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    // one set of side stream + events per device, held across fork -> launches -> join (everything below only enqueues)
    std::lock_guard<std::mutex> lk(g_stage_side_mu);
    StageSide* side = nullptr;
    for (auto& m : g_stage_side) if (m.device == dev) side = &m;
    if (!side) {
        // built completely before it is registered; a failure half way destroys what exists (no stream or event is leaked)
        StageSide m;
        m.device = dev;
        hipError_t e = hipStreamCreateWithFlags(&m.s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m.fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m.join, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (m.join) (void)hipEventDestroy(m.join);
            if (m.fork) (void)hipEventDestroy(m.fork);
            if (m.s) (void)hipStreamDestroy(m.s);
            return ::porla::hip_fail(e, "crebuild stage: side stream / events", __FILE__, __LINE__);
        }
        g_stage_side.push_back(m);
        side = &g_stage_side.back();
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    PORLA_HIP(hipEventRecord(side->fork, stream));
    PORLA_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
    // From here on the side stream may hold work that writes d_macs_x / d_macs_y: EVERY exit joins it back into hip_stream, also the
    // failing ones -- a caller that gets an error may free its buffers as soon as hip_stream has drained, and the next call's
    // fork / join records must not interleave with a stage still running
    rc = porla_icc_mac_encode_xy_device(d_macs_in, n_rows, 0, write_step, d_macs_x, d_macs_y, side->s);
    hipError_t je = hipEventRecord(side->join, side->s);
    uint8_t* sc = (uint8_t*)d_scalars_xy;
    if (!rc)
        rc = porla_icc_encode_xy_device(d_rows_in, n_rows, n_cols, 0, write_step, nullptr, d_aligned_x, sc, nullptr, d_aligned_y,
                                        sc + 32 * n_rows * n_cols, 0, stream);
    // both parts' alignment scalars lie back to back: ONE batch of 2 n rows
    if (!rc) rc = commit_rows(sc, true, 2 * n_rows, n_cols, (uint8_t*)d_commits_xy, stream, /*guest_room=*/true);
    if (je == hipSuccess) je = hipStreamWaitEvent(stream, side->join, 0);
    if (je != hipSuccess) {
        // the join itself failed: fall back to a host wait so that no side work outlives the call
        (void)hipStreamSynchronize(side->s);
        if (!rc) rc = ::porla::hip_fail(je, "crebuild stage: join of the side stream", __FILE__, __LINE__);
    }
    return rc;
}

// rows resident on the device, results wanted on the host NOW (the audit's align_MAC commitment, Server.hpp:903 -> :550-560, on the
// scalars porla_audit_combine_device left in HBM): up to 64 rows go through the single-launch kernel on `hip_stream` -- behind
// whatever produced the rows there -- and the host polls the pinned result; more rows: the batch kernels and one copy back
int porla_kzg_commit_batch_device_to_host(const void* d_rows, size_t n_rows, uint8_t* out, void* hip_stream) {
    if (n_rows && (!d_rows || !out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n_rows == 0) return PORLA_OK;
    const size_t len = kzg_n_samples();
    hipStream_t stream = (hipStream_t)hip_stream;
    if (FixedBase<Bn254G1>::small_ok(n_rows, len)) {
        std::unique_lock<std::mutex> lk(g.mu);
        KzgState::Dev* kd = nullptr;
        if ((rc = refresh_srs_locked(&kd))) {
            if (rc == PORLA_ERR_STATE) set_last_error("porla: SRS not initialised (call init_SRS / init_SRS_from_data first)");
            return rc;
        }
        std::unique_lock<std::mutex> lkfb(kd->fb.mu);
        lk.unlock();
        uint8_t* op[FB_SMALL_MAX_ROWS];
        for (size_t r = 0; r < n_rows; r++) op[r] = out + 64 * r;
        return kd->fb.commit_small(nullptr, n_rows, len, op, stream, (const uint8_t*)d_rows);
    }
    void* d_out = nullptr;
    PORLA_HIP(hipMalloc(&d_out, n_rows * 64));
    rc = commit_rows((const uint8_t*)d_rows, true, n_rows, len, (uint8_t*)d_out, stream);
    hipError_t e = rc ? hipSuccess : hipMemcpyAsync(out, d_out, n_rows * 64, hipMemcpyDeviceToHost, stream);
    hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
    if (e2 != hipSuccess) return hip_fail(e2, "hipStreamSynchronize", __FILE__, __LINE__);
    return PORLA_OK;
}
// Server::audit for the KZG build in ONE call (Server.hpp:564-931 after the challenge has been drawn), everything resident in HBM:
//   the two MSMs over the challenged MACs start first, on the audit slot's own stream (msm_pair_gather_begin);
//   meanwhile: row combine + alignment scalars (audit.hip) -> B and c land in pinned host memory; y = B(z) and the quotient h on
//   the host; ONE three-row launch commits c (align_MAC, :903 -> :550-560), B and h (create_proof, :907 -> main.go:153-175);
//   then the MSM pair is collected.
namespace {
struct AuditPinned { int device = -1; uint8_t* h = nullptr; size_t cap = 0; };
std::mutex g_audit_call_mu;           // one audit at a time per process (the pinned staging and the audit slot are its own)
std::vector<AuditPinned> g_audit_pinned;
}  // namespace
int porla_kzg_audit_device(const void* d_rows64, const uint64_t* d_idx64, const uint32_t* d_coef64, size_t n64, const void* d_rows32,
                           const uint64_t* d_idx32, const uint32_t* d_coef32, size_t n32, const void* d_mac_store,
                           const void* d_align_store, const uint64_t* d_mac_idx, const uint32_t* d_mac_coef, size_t n_macs,
                           unsigned long long random_point, uint8_t combined_mac[64], uint8_t combined_align[64],
                           uint8_t align_value[64], uint8_t commitment[64], uint8_t proof_h[64], uint8_t proof_point[32],
                           uint8_t proof_claim[32], uint8_t* b_out, void* hip_stream) {
    if (!combined_mac || !combined_align || !align_value || !commitment || !proof_h || !proof_point || !proof_claim ||
        (n_macs && (!d_mac_store || !d_align_store || !d_mac_idx || !d_mac_coef))) {
        set_last_error("porla: null argument");
        return PORLA_ERR_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    const size_t n = kzg_n_samples();
    if (n == 0) { set_last_error("porla: SRS not initialised (call init_SRS / init_SRS_from_data first)"); return PORLA_ERR_STATE; }
    std::lock_guard<std::mutex> lk(g_audit_call_mu);
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    AuditPinned* pin = nullptr;
    for (auto& p : g_audit_pinned) if (p.device == dev) pin = &p;
    if (!pin) { g_audit_pinned.push_back(AuditPinned()); pin = &g_audit_pinned.back(); pin->device = dev; }
    if (pin->cap < 64 * n) {
        if (pin->h) PORLA_HIP(hipHostFree(pin->h));
        pin->h = nullptr; pin->cap = 0;
        PORLA_HIP(hipHostMalloc((void**)&pin->h, 64 * n, hipHostMallocMapped | hipHostMallocCoherent));
        pin->cap = 64 * n;
    }
    void* pin_dev = nullptr;
    PORLA_HIP(hipHostGetDevicePointer(&pin_dev, pin->h, 0));
    uint8_t* h_b = pin->h;                  // B mod p_icc, n 32-byte big-endian values
    uint8_t* h_c = pin->h + 32 * n;         // the alignment scalars
    // hip_stream orders the INPUTS: the combine runs on it as given (NULL = the null stream), the pair on the audit slot's own
    // stream behind an event recorded on hip_stream now -- index / coefficient arrays the caller has just uploaded asynchronously
    // on it are complete before any kernel of the audit reads them (one record + one wait: ~4 us of a 150 us call)
    hipStream_t stream = (hipStream_t)hip_stream;
    Workspace* aw = nullptr;
    if ((rc = get_workspace_slot(MSM_AUDIT_SLOT, &aw))) return rc;
    if ((rc = order_after_caller(aw, (hipStream_t)hip_stream, stream, aw->own_stream))) return rc;
    const bool pair = n_macs >= 1 && n_macs <= 32768;
    bool pair_begun = false;
    auto collect_pair = [&]() -> int {
        if (!pair_begun) return PORLA_OK;
        XYZZ<Fp> ta, tb;
        int r2 = msm_pair_end<Bn254G1>(MSM_AUDIT_SLOT, &ta, &tb);
        if (r2) return r2;
        const XYZZ<Fp> both[2] = {ta, tb};
        Affine<Fp> aff[2];
        h_batch_xyzz_to_affine64<Fp>(both, 2, aff);          // one inversion for the two sums
        h_affine_to_bytes<Fp>(combined_mac, aff[0]);
        h_affine_to_bytes<Fp>(combined_align, aff[1]);
        return PORLA_OK;
    };
    // the combine is enqueued FIRST: its two short kernels take their compute units before the pair's 256 long-lived blocks do (begun
    // the other way round the combine was seen to wait ~85 us behind them), then the pair starts on the audit slot's own stream
    rc = porla_audit_combine_device(d_rows64, d_idx64, d_coef64, n64, d_rows32, d_idx32, d_coef32, n32, n, 0, nullptr, nullptr,
                                    pin_dev, (uint8_t*)pin_dev + 32 * n, stream);
    if (rc == PORLA_OK && pair) {
        rc = msm_pair_gather_begin<Bn254G1>(MSM_AUDIT_SLOT, (const uint8_t*)d_mac_store, (const uint8_t*)d_align_store, d_mac_idx,
                                                d_mac_coef, n_macs, aw->own_stream);
        pair_begun = rc == PORLA_OK;
    }
    if (rc == PORLA_OK && hipStreamSynchronize(stream) != hipSuccess) {
        set_last_error("porla: hipStreamSynchronize failed in the audit");
        rc = PORLA_ERR_HIP;
    }
    if (rc) { (void)collect_pair(); return rc; }
    std::vector<uint8_t> three(3 * 32 * n);
    memcpy(three.data(), h_c, 32 * n);
    memcpy(three.data() + 32 * n, h_b, 32 * n);
    kzg_open_rows(h_b, n, random_point, three.data() + 64 * n, proof_point, proof_claim);
    if (b_out) memcpy(b_out, h_b, 32 * n);
    uint8_t outs[192];
    // the three row sums stay projective until the pair's two sums are in: ONE inversion normalises all five points
    XYZZ<Fp> five[5];
    bool raw3 = false;
    if (pair_begun && FixedBase<Bn254G1>::small_ok(3, n)) {
        std::unique_lock<std::mutex> lks(g.mu);
        KzgState::Dev* kd = nullptr;
        rc = refresh_srs_locked(&kd);
        if (rc == PORLA_OK) {
            std::unique_lock<std::mutex> lkfb(kd->fb.mu);
            lks.unlock();
            const uint8_t* rp[3] = {three.data(), three.data() + 32 * n, three.data() + 64 * n};
            rc = kd->fb.commit_small(rp, 3, n, nullptr, engine_stream(), nullptr, five);
            raw3 = rc == PORLA_OK;
        }
    } else {
        rc = commit_rows(three.data(), false, 3, n, outs, nullptr);
    }
    int rc2;
    if (raw3) {
        rc2 = msm_pair_end<Bn254G1>(MSM_AUDIT_SLOT, &five[3], &five[4]);
        pair_begun = false;
        if (rc2 == PORLA_OK) {
            Affine<Fp> aff[5];
            h_batch_xyzz_to_affine64<Fp>(five, 5, aff);
            for (int i = 0; i < 3; i++) h_affine_to_bytes<Fp>(outs + 64 * i, aff[i]);
            h_affine_to_bytes<Fp>(combined_mac, aff[3]);
            h_affine_to_bytes<Fp>(combined_align, aff[4]);
        }
    } else {
        rc2 = collect_pair();
    }
    if (rc) return rc;
    if (rc2) return rc2;
    if (!pair) {
        // more challenged rows than the single-launch pair takes (or none): the blocking pair form
        if ((rc = porla_bn254_audit_msm_pair_device(d_mac_store, d_align_store, d_mac_idx, d_mac_coef, n_macs, combined_mac, combined_align, stream)))
            return rc;
    }
    memcpy(align_value, outs, 64);
    memcpy(commitment, outs + 64, 64);
    memcpy(proof_h, outs + 128, 64);
    return PORLA_OK;
}
int porla_kzg_commit_batch_host(const uint8_t* rows, size_t n_rows, uint8_t* out) {
    if (n_rows && (!rows || !out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    return commit_rows(rows, false, n_rows, kzg_n_samples(), out, nullptr);
}
// Server::HAdd for the KZG build, everything it computes before the level bookkeeping (Server.hpp:1388-1428): data_B2 = data * wt
// aligned mod p_icc, MAC_B2 = wt * MAC, MAC_align_B2 = Commit(alignment scalars of data_B2) -- align_MAC's compute_digest_from_srs
// (Server.hpp:531-560) on the scalars the device derived.  n_cols = NUM_CHUNKS = the SRS size.
int porla_kzg_hadd_host(const uint8_t* data_in, const uint8_t mac_in[64], size_t n_total, unsigned long long write_step,
                        uint8_t* data_b2_out, uint8_t mac_b2_out[64], uint8_t mac_align_b2_out[64]) {
    if (!data_in || !mac_in || !data_b2_out || !mac_b2_out || !mac_align_b2_out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    const size_t n_cols = kzg_n_samples();
    if (n_cols == 0) { set_last_error("porla: SRS not initialised"); return PORLA_ERR_STATE; }
    std::vector<uint8_t> scalars(32 * n_cols);
    uint8_t wt[32];
    int rc = porla_icc_hadd_host(data_in, n_cols, n_total, write_step, 0, data_b2_out, scalars.data(), 0, wt);
    if (rc) return rc;
    if ((rc = porla_icc_mac_scale_host(mac_in, n_total, write_step, 0, mac_b2_out))) return rc;
    return commit_coalesced(scalars.data(), mac_align_b2_out);      // infinity + Commit(c) (bn254_add(B, align_value), B = infinity)
}

// rows are independent: device g of `devices` commits the row range [g R / G, (g+1) R / G) from its own host thread against its
// own resident copy of the SRS table; the results land in the caller's `out`, nothing is exchanged
int porla_kzg_commit_batch_host_multi(const uint8_t* rows, size_t n_rows, uint8_t* out, int devices) {
    if (n_rows && (!rows || !out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    int visible = 0, first = 0;
    PORLA_HIP(hipGetDeviceCount(&visible));
    PORLA_HIP(hipGetDevice(&first));
    int G = devices <= 0 ? visible : (devices < visible ? devices : visible);
    if ((size_t)G > n_rows) G = (int)n_rows;
    if (G < 1) G = 1;
    const size_t len = kzg_n_samples();
    std::vector<int> rcs((size_t)G, PORLA_OK);
    std::vector<std::string> errs((size_t)G);
    auto worker = [&](int d) {
        if (hipSetDevice((first + d) % visible) != hipSuccess) { rcs[d] = PORLA_ERR_HIP; errs[d] = "porla: hipSetDevice failed"; return; }
        size_t lo, hi;
        porla_shard_range(n_rows, d, G, &lo, &hi);
        rcs[d] = commit_rows(rows + lo * len * 32, false, hi - lo, len, out + 64 * lo, nullptr);
        if (rcs[d]) errs[d] = porla_gpu_last_error();
    };
    std::vector<std::thread> th;
    for (int d = 1; d < G; d++) th.emplace_back(worker, d);
    worker(0);
    for (auto& t : th) t.join();
    if (G > 1) (void)hipSetDevice(first);
    for (int d = 0; d < G; d++) if (rcs[d]) { set_last_error(errs[d]); return rcs[d]; }
    return PORLA_OK;
}
int porla_kzg_set_commit_window(int window_bits) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (window_bits != g.commit_window) { g.commit_window = window_bits; g.version++; }
    return PORLA_OK;
}
// frees the HBM copies that belong to the KZG state (SRS, its window-multiples table -- 56 GB by default --, the one-point
// tables of the client-side batches and scratch); they are rebuilt by the next call that needs them
int porla_kzg_release_device_memory(void) {
    // lock order as everywhere: the state, then the tables -- a commit that has dropped g.mu still holds its table's mutex
    // (compute_digest_from_srs is called from 8 pool threads, Server.hpp:550-560)
    // the host batches' staging buffers first, outside g.mu: a host batch holds its staging mutex while the device entry it calls
    // takes g.mu
    for (ClientIo& io : g_client_io) {
        std::lock_guard<std::mutex> lio(io.mu);
        if (io.d) (void)hipFree(io.d);
        io.d = nullptr; io.cap = 0;
    }
    std::lock_guard<std::mutex> lk(g.mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (KzgState::Dev* kd : g.devs) {
        if (!kd) continue;
        std::lock_guard<std::mutex> l1(kd->fb.mu), l2(kd->fb_g.mu), l3(kd->fb_h.mu), l4(kd->fb_gh.mu);
        (void)hipSetDevice(kd->device);
        kd->fb.release();
        kd->fb_g.release();
        kd->fb_h.release();
        kd->fb_gh.release();
        if (kd->d_srs) (void)hipFree(kd->d_srs);
        kd->d_srs = nullptr; kd->d_srs_cap = 0;
        if (kd->d_eval) (void)hipFree(kd->d_eval);
        kd->d_eval = nullptr; kd->d_eval_cap = 0;
        if (kd->d_tau29) (void)hipFree(kd->d_tau29);
        kd->d_tau29 = nullptr; kd->tau29_n = 0;
        kd->srs_version = kd->g_version = kd->h_version = kd->gh_version = 0;
    }
    (void)hipSetDevice(cur);
    return PORLA_OK;
}

int porla_kzg_commit_shape(int* window_bits, int* windows) {
    std::lock_guard<std::mutex> lk(g.mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const KzgState::Dev* kd = dev >= 0 && dev < 16 ? g.devs[dev] : nullptr;
    if (window_bits) *window_bits = kd ? kd->fb.c : 0;
    if (windows) *windows = kd ? kd->fb.W : 0;
    return PORLA_OK;
}

}  // extern "C"
