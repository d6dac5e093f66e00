// affine_batch_ubench.hip -- what batched-affine (Montgomery's trick) bucket accumulation could reach on gfx950, measured in
// its BEST case, against the mixed addition k_bucket_sum30 runs now.  (VERDICT r02 "next" item 1; the add it would replace:
// porla/Utils/secp256k1_lib/ecmult_impl.h:520-542 / group_impl.h:389-436 on the reference side, ec30.hip.h:xyzz30_madd here.)
//
// An affine addition P + Q costs 1/(x2 - x1), i.e. an inversion; Montgomery's trick shares ONE inversion over k independent
// additions at 3 products each:  forward  prefix_j = prefix_(j-1) * d_j  (d_j = x2_j - x1_j; prefix_(j-1) kept),
// one inversion of prefix_k, backward  1/d_j = inv * prefix_(j-1), inv *= d_j, then lambda = (y2 - y1) / d_j,
// x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1: 5 products + 1 square + 1/k of an inversion against 8 products +
// 2 squares.  On a GPU the inversion (inv30.hip.h: ~22 k instructions) is executed by all 64 lanes of a wave at once, so k
// is a PER-LANE batch, and the k operand pairs and prefix products of a lane live in memory between the passes:
// registers hold ~3 additions' worth, LDS 160 KB / 224 B = 730 per CU.
//
// Kernels (BN254 base field, 9 x 30-bit limbs, the products of fe30.hip.h):
//   k_xyzz_stream   the present form: a lane folds 2k points into one XYZZ accumulator in registers, points STREAMED
//                   ([j][lane] layout, 64 coalesced bytes per addition) -- k_bucket_sum30 without its gathers
//   k_affine_batch  a lane adds k pairs (x1, y1) + (x2, y2): forward pass (reads x1, x2: 64 B; writes the prefix: 32 B),
//                   one division-step inversion, backward pass (reads the prefix 32 B and both points 128 B, writes the
//                   sum 64 B): 320 B per addition, every access coalesced -- no bucket structure, no gathers, no second
//                   round: the ceiling of ANY batched-affine accumulation
//   k_inv_only      the inversions alone
// Every result of k_affine_batch is checked against the XYZZ addition of the same pair on the host side of this file's device
// code (k_check).  Output: G additions/s per kernel and k; run under rocprofv3 --pmc for the instruction and byte counters
// (tools/affine_batch_profile.sh -> profiles/r03_d_*).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ec30.hip.h"
#include "inv30.hip.h"

using namespace porla;

// the plain mixed addition (the library keeps only the sign-alternating run form since round 5): one addition, sign restored
template <class M>
__device__ __forceinline__ void xyzz30_madd(XYZZ30<M>& p, const F30<M>& ax, const F30<M>& ay) {
    bool flip = false;
    xyzz30_madd_flip<M>(p, flip, ax, ay);
    xyzz30_flip_finish<M>(p, flip);
}
using M = Bn254Fp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ F30<M> ld30(const uint32_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    uint32_t t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return f30_unpack<M>(t);
}
__device__ __forceinline__ void st30(uint32_t* p, const F30<M>& v) {   // v: normal limbs, value < 2^256
    uint32_t t[8];
    f30_pack<M>(t, v);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(t[0], t[1], t[2], t[3]);
    q[1] = make_uint4(t[4], t[5], t[6], t[7]);
}

// operand arrays: coordinate c (0 = x1, 1 = y1, 2 = x2, 3 = y2) of pair j of lane t at  ops + ((c * k + j) * lanes + t) * 8
// words: a wave reads 64 x 32 contiguous bytes per coordinate.  Values: canonical residues in the 2^270 form.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_xyzz_stream(const uint32_t* __restrict__ ops, uint32_t k, uint32_t lanes, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ30<M> acc;
    acc.inf = true;
    for (uint32_t j = 0; j < k; j++) {
        for (int h = 0; h < 2; h++) {
            const F30<M> x = ld30(ops + ((size_t)((2 * h) * k + j) * lanes + t) * 8);
            const F30<M> y = ld30(ops + ((size_t)((2 * h + 1) * k + j) * lanes + t) * 8);
            xyzz30_madd<M>(acc, x, y);
        }
    }
    XYZZ<M>* dst = reinterpret_cast<XYZZ<M>*>(out) + t;
    xyzz30_store_lazy<M>(dst, acc);
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_affine_batch(const uint32_t* __restrict__ ops, uint32_t k, uint32_t lanes, uint32_t* __restrict__ prefix,
               uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    // ---- forward: prefix products of the denominators
    F30<M> acc = f30_const<M>(M::R1_30);
    for (uint32_t j = 0; j < k; j++) {
        const F30<M> x1 = ld30(ops + ((size_t)(0 * k + j) * lanes + t) * 8);
        const F30<M> x2 = ld30(ops + ((size_t)(2 * k + j) * lanes + t) * 8);
        st30(prefix + ((size_t)j * lanes + t) * 8, acc);
        acc = f30_mul<M>(acc, f30_sub<M, 2>(x2, x1));
    }
    // ---- one inversion per lane: (T 2^270)^-1 as an integer, times 2^810 / 2^270 = T^-1 2^270
    F30<M> inv;
    {
        Fe<M> tot;
        f30_pack<M>(tot.v, acc);                                           // < p + 2^246 < 2^256
        const F30<M> rr = f30_const<M>(M::RR_30);
        inv = f30_mul<M>(f30_inv_safegcd_raw<M>(tot), f30_mul<M>(rr, rr));
    }
    // ---- backward: the k additions
    for (uint32_t j = k; j-- > 0;) {
        const F30<M> x1 = ld30(ops + ((size_t)(0 * k + j) * lanes + t) * 8);
        const F30<M> y1 = ld30(ops + ((size_t)(1 * k + j) * lanes + t) * 8);
        const F30<M> x2 = ld30(ops + ((size_t)(2 * k + j) * lanes + t) * 8);
        const F30<M> y2 = ld30(ops + ((size_t)(3 * k + j) * lanes + t) * 8);
        const F30<M> pre = ld30(prefix + ((size_t)j * lanes + t) * 8);
        const F30<M> d = f30_sub<M, 2>(x2, x1);
        const F30<M> dinv = f30_mul<M>(inv, pre);                          // 1 / d_j
        inv = f30_mul<M>(inv, d);
        const F30<M> lam = f30_mul<M>(f30_sub<M, 2>(y2, y1), dinv);
        const F30<M> ll = f30_sqr<M>(lam);
        F30<M> sx;                                                         // x1 + x2 (< 2 p), normal limbs
#pragma unroll
        for (int i = 0; i < 9; i++) sx.v[i] = x1.v[i] + x2.v[i];
        f30_ripple<M>(sx);
        const F30<M> x3 = f30_sub<M, 3>(ll, sx);                           // <= 4 p
        const F30<M> y3 = f30_sub<M, 2>(f30_mul<M>(lam, f30_sub<M, 6>(x1, x3)), y1);   // x3 <= 4 p + 2^246: K = 6; y3 <= 3 p
        st30(out + ((size_t)(2 * j) * lanes + t) * 8, x3);
        st30(out + ((size_t)(2 * j + 1) * lanes + t) * 8, y3);
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_inv_only(const uint32_t* __restrict__ ops, uint32_t lanes, uint32_t reps, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F30<M> v = ld30(ops + (size_t)t * 8);
    for (uint32_t r = 0; r < reps; r++) {
        Fe<M> a;
        f30_pack<M>(a.v, v);
        a.v[0] |= 1u;
        v = f30_inv_safegcd_raw<M>(a);
        v.v[8] &= 0xffffu;
    }
    st30(out + (size_t)t * 8, v);
}

// pair j of lane t once more as an XYZZ addition; compares the affine images.  bad += 1 per mismatch.
__global__ void k_check(const uint32_t* __restrict__ ops, uint32_t k, uint32_t lanes, const uint32_t* __restrict__ got,
                        uint32_t* __restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    for (uint32_t j = 0; j < k; j += (k > 8 ? k / 8 : 1)) {
        XYZZ30<M> acc;
        acc.inf = true;
        xyzz30_madd<M>(acc, ld30(ops + ((size_t)(0 * k + j) * lanes + t) * 8), ld30(ops + ((size_t)(1 * k + j) * lanes + t) * 8));
        xyzz30_madd<M>(acc, ld30(ops + ((size_t)(2 * k + j) * lanes + t) * 8), ld30(ops + ((size_t)(3 * k + j) * lanes + t) * 8));
        // affine x = X / ZZ, y = Y / ZZZ  <=>  x3 ZZ == X and y3 ZZZ == Y (all in the 2^270 form)
        const F30<M> x3 = ld30(got + ((size_t)(2 * j) * lanes + t) * 8), y3 = ld30(got + ((size_t)(2 * j + 1) * lanes + t) * 8);
        const Fe<M> lx = f30_to_fe_canonical<M>(f30_mul<M>(x3, acc.zz)), rx = f30_to_fe_canonical<M>(f30_mul<M>(acc.x, f30_const<M>(M::R1_30)));
        const Fe<M> ly = f30_to_fe_canonical<M>(f30_mul<M>(y3, acc.zzz)), ry = f30_to_fe_canonical<M>(f30_mul<M>(acc.y, f30_const<M>(M::R1_30)));
        bool ok = !acc.inf;
        for (int i = 0; i < 8; i++) ok = ok && lx.v[i] == rx.v[i] && ly.v[i] == ry.v[i];
        if (!ok) atomicAdd(bad, 1u);
    }
}

// points on the curve for the check to mean something: lane t, slot s gets (s * lanes + t + 2) * G by repeated addition is
// too slow here; instead the operands are ANY field elements -- the affine addition law and the XYZZ law are the same
// rational maps in (x1, y1, x2, y2) whether or not the points lie on the curve (neither uses the curve equation when
// x1 != x2), which is all the cross-check needs.
__global__ void k_fill(uint32_t* __restrict__ ops, size_t values) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= values) return;
    uint32_t w[8];
    uint64_t s = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    for (int q = 0; q < 8; q++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[q] = (uint32_t)(s >> 16); }
    w[7] &= 0x1fffffffu;                                                   // < 2^253 < p: a canonical residue
    uint4* d = reinterpret_cast<uint4*>(ops + i * 8);
    d[0] = make_uint4(w[0], w[1], w[2], w[3]);
    d[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <class F>
static double time_ms(F f, int reps = 3) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    const uint32_t lanes = (uint32_t)CUS * 4 * 256;                         // 4 waves per SIMD, one round
    const uint32_t kmax = argc > 1 ? (uint32_t)atoi(argv[1]) : 128;
    uint32_t *ops, *prefix, *out, *bad;
    const size_t values = (size_t)4 * kmax * lanes;
    CK(hipMalloc(&ops, values * 32));
    CK(hipMalloc(&prefix, (size_t)kmax * lanes * 32));
    CK(hipMalloc(&out, (size_t)2 * kmax * lanes * 32 + (size_t)lanes * 128));
    CK(hipMalloc(&bad, 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((values + 255) / 256)), dim3(256), 0, 0, ops, values);
    CK(hipDeviceSynchronize());
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"lanes\": %u, \"rows\": [\n", prop.gcnArchName, CUS, lanes);
    bool first = true;
    for (uint32_t k = 8; k <= kmax; k *= 2) {
        const double adds_x = 2.0 * k * lanes, adds_a = (double)k * lanes;
        const double ms_x = time_ms([&] { hipLaunchKernelGGL(k_xyzz_stream, dim3(lanes / 256), dim3(256), 0, 0, ops, k, lanes, out); });
        const double ms_a = time_ms([&] { hipLaunchKernelGGL(k_affine_batch, dim3(lanes / 256), dim3(256), 0, 0, ops, k, lanes, prefix, out); });
        CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_check, dim3(lanes / 256), dim3(256), 0, 0, ops, k, lanes, out, bad);
        uint32_t nbad = 0;
        CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
        printf("%s  {\"k\": %u, \"xyzz_stream_Gadd_s\": %.2f, \"xyzz_ms\": %.3f, \"affine_batch_Gadd_s\": %.2f, \"affine_ms\": %.3f, "
               "\"affine_bytes_per_add\": 320, \"affine_TB_s\": %.2f, \"mismatches\": %u}",
               first ? "" : ",\n", k, adds_x / ms_x / 1e6, ms_x, adds_a / ms_a / 1e6, ms_a, adds_a * 320 / ms_a / 1e9, nbad);
        first = false;
        fflush(stdout);
    }
    const uint32_t reps = 4;
    const double ms_i = time_ms([&] { hipLaunchKernelGGL(k_inv_only, dim3(lanes / 256), dim3(256), 0, 0, ops, lanes, reps, out); });
    printf("\n], \"inversions_G_s\": %.3f, \"inversion_us_per_wave\": %.1f}\n", (double)lanes * reps / ms_i / 1e6, ms_i * 1e3 / reps);
    return 0;
}
