// Instruction-throughput micro-benchmarks for the integer pipeline of gfx950 (run on the GPU box):
// how many 32x32->64 multiply-adds, 64-bit FMAs and field products per second the chip sustains.
// These numbers calibrate the VALU-side model in DESIGN.md next to the mandated HBM roofline.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -I porla_amd/csrc -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fe.hip.h"
#include "ec.hip.h"
using namespace porla;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ void k_op(uint32_t* out, int iters, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a0 = tid + seed, a1 = tid * 3 + 1, a2 = tid * 5 + 2, a3 = tid * 7 + 3, a4 = tid ^ 0x55, a5 = tid + 99, a6 = tid * 11, a7 = ~tid;
    uint32_t x = tid * 2654435761u + seed, y = seed ^ 0x9e3779b9u;
    double d0 = tid, d1 = tid + 1, d2 = tid + 2, d3 = tid + 3, d4 = 1.5, d5 = 2.5, d6 = 3.5, d7 = 4.5, dx = 1.0000001, dy = 0.5;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) {  // v_mad_u64_u32, 8 independent chains
            a0 = (uint64_t)x * (uint32_t)a0 + a0; a1 = (uint64_t)x * (uint32_t)a1 + a1; a2 = (uint64_t)x * (uint32_t)a2 + a2; a3 = (uint64_t)x * (uint32_t)a3 + a3;
            a4 = (uint64_t)y * (uint32_t)a4 + a4; a5 = (uint64_t)y * (uint32_t)a5 + a5; a6 = (uint64_t)y * (uint32_t)a6 + a6; a7 = (uint64_t)y * (uint32_t)a7 + a7;
        } else if (OP == 1) {  // v_mul_lo_u32
            uint32_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
            b0 *= x; b1 *= x; b2 *= x; b3 *= x; b4 *= y; b5 *= y; b6 *= y; b7 *= y;
            a0 = b0 | 1; a1 = b1 | 1; a2 = b2 | 1; a3 = b3 | 1; a4 = b4 | 1; a5 = b5 | 1; a6 = b6 | 1; a7 = b7 | 1;
        } else if (OP == 2) {  // v_fma_f64
            d0 = fma(d0, dx, dy); d1 = fma(d1, dx, dy); d2 = fma(d2, dx, dy); d3 = fma(d3, dx, dy);
            d4 = fma(d4, dx, dy); d5 = fma(d5, dx, dy); d6 = fma(d6, dx, dy); d7 = fma(d7, dx, dy);
        } else if (OP == 3) {  // v_mul_hi_u32
            uint32_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
            b0 = __umulhi(b0, x) + 3; b1 = __umulhi(b1, x) + 5; b2 = __umulhi(b2, x) + 7; b3 = __umulhi(b3, x) + 9;
            b4 = __umulhi(b4, y) + 11; b5 = __umulhi(b5, y) + 13; b6 = __umulhi(b6, y) + 15; b7 = __umulhi(b7, y) + 17;
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (OP == 4) {  // 32-bit add (full rate reference)
            uint32_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
            b0 += x ^ b1; b1 += x ^ b2; b2 += x ^ b3; b3 += x ^ b4; b4 += y ^ b5; b5 += y ^ b6; b6 += y ^ b7; b7 += y ^ b0;
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (OP == 5) {  // v_mad_u32_u24
            uint32_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
            b0 = __umul24(b0, x) + b1; b1 = __umul24(b1, x) + b2; b2 = __umul24(b2, x) + b3; b3 = __umul24(b3, x) + b4;
            b4 = __umul24(b4, y) + b5; b5 = __umul24(b5, y) + b6; b6 = __umul24(b6, y) + b7; b7 = __umul24(b7, y) + b0;
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        }
    }
    uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    double dr = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    out[tid] = (uint32_t)r ^ (uint32_t)(r >> 32) ^ (uint32_t)dr;
}

template <class M>
__global__ void k_femul(uint32_t* io, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fe<M> x, y;
    for (int i = 0; i < 8; i++) { x.v[i] = io[tid * 8 + i]; y.v[i] = io[tid * 8 + i] ^ 0x01010101u; }
    x.v[7] &= 0x0fffffff; y.v[7] &= 0x0fffffff;
    for (int i = 0; i < iters; i++) { x = fe_mul<M>(x, y); y = fe_mul<M>(y, x); }
    for (int i = 0; i < 8; i++) io[tid * 8 + i] = x.v[i] ^ y.v[i];
}
template <class M>
__global__ void k_madd(uint32_t* io, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Affine<M> a;
    for (int i = 0; i < 8; i++) { a.x.v[i] = io[tid * 8 + i]; a.y.v[i] = io[tid * 8 + i] ^ 0x01010101u; }
    a.x.v[7] &= 0x0fffffff; a.y.v[7] &= 0x0fffffff;
    XYZZ<M> p = xyzz_from_affine<M>(a);
    a.x.v[0] ^= 5;
    for (int i = 0; i < iters; i++) { xyzz_madd<M>(p, a); a.x.v[1] += 1; }
    for (int i = 0; i < 8; i++) io[tid * 8 + i] = p.x.v[i] ^ p.y.v[i] ^ p.zz.v[i] ^ p.zzz.v[i];
}

template <class F>
static double time_ms(F f, int reps = 3) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz  arch=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.gcnArchName);
    const int CUS = prop.multiProcessorCount;
    uint32_t* buf; CK(hipMalloc(&buf, (size_t)CUS * 2048 * 8 * 4 * 2));
    std::vector<uint32_t> h((size_t)CUS * 2048 * 8 * 2);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345);
    CK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_fma_f64", "v_mul_hi_u32", "v_add_u32(+xor)", "v_mad_u32_u24"};
    const int iters = 4096;
    for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2) {
        int blocks = CUS * wavesPerSimd;  // 256 threads = 4 waves = 1 per SIMD
        double ms;
        ms = time_ms([&] { hipLaunchKernelGGL(k_op<0>, dim3(blocks), dim3(256), 0, 0, buf, iters, 1u); });
        printf("%-16s waves/SIMD=%d  %.3f ms  %.2f Tops/s\n", names[0], wavesPerSimd, ms, (double)blocks * 256 * iters * 8 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_op<1>, dim3(blocks), dim3(256), 0, 0, buf, iters, 1u); });
        printf("%-16s waves/SIMD=%d  %.3f ms  %.2f Tops/s\n", names[1], wavesPerSimd, ms, (double)blocks * 256 * iters * 8 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_op<2>, dim3(blocks), dim3(256), 0, 0, buf, iters, 1u); });
        printf("%-16s waves/SIMD=%d  %.3f ms  %.2f Tops/s\n", names[2], wavesPerSimd, ms, (double)blocks * 256 * iters * 8 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_op<3>, dim3(blocks), dim3(256), 0, 0, buf, iters, 1u); });
        printf("%-16s waves/SIMD=%d  %.3f ms  %.2f Tops/s\n", names[3], wavesPerSimd, ms, (double)blocks * 256 * iters * 8 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_op<4>, dim3(blocks), dim3(256), 0, 0, buf, iters, 1u); });
        printf("%-16s waves/SIMD=%d  %.3f ms  %.2f Tops/s (x2 ops)\n", names[4], wavesPerSimd, ms, (double)blocks * 256 * iters * 8 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_op<5>, dim3(blocks), dim3(256), 0, 0, buf, iters, 1u); });
        printf("%-16s waves/SIMD=%d  %.3f ms  %.2f Tops/s\n", names[5], wavesPerSimd, ms, (double)blocks * 256 * iters * 8 / ms / 1e9);
    }
    for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2) {
        int blocks = CUS * wavesPerSimd;
        const int it = 512;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_femul<Bn254Fp>, dim3(blocks), dim3(256), 0, 0, buf, it); });
        printf("fe_mul<Bn254Fp>   waves/SIMD=%d  %.3f ms  %.2f Gmul/s  (%.0f cycles/mul/wave @2.4GHz)\n", wavesPerSimd, ms,
               (double)blocks * 256 * it * 2 / ms / 1e6, ms * 1e-3 * 2.4e9 / (it * 2) / wavesPerSimd);
        ms = time_ms([&] { hipLaunchKernelGGL(k_femul<Secp256k1Fp>, dim3(blocks), dim3(256), 0, 0, buf, it); });
        printf("fe_mul<Secp256k1> waves/SIMD=%d  %.3f ms  %.2f Gmul/s\n", wavesPerSimd, ms, (double)blocks * 256 * it * 2 / ms / 1e6);
        if (wavesPerSimd <= 4) {
            ms = time_ms([&] { hipLaunchKernelGGL(k_madd<Bn254Fp>, dim3(blocks), dim3(256), 0, 0, buf, 128); });
            printf("xyzz_madd<Bn254>  waves/SIMD=%d  %.3f ms  %.2f Gadd/s\n", wavesPerSimd, ms, (double)blocks * 256 * 128 / ms / 1e6);
        }
    }
    CK(hipFree(buf));
    return 0;
}
