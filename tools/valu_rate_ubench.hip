// valu_rate_ubench.hip -- issue cost of the integer VALU instructions the field arithmetic is made of, in cycles per wave
// instruction per SIMD, at 1 / 2 / 4 / 8 waves per SIMD: decides what an instruction saved is worth (is a mask or a shift as
// expensive as a v_mad_u64_u32?).  Independent instructions (8 accumulator chains), 64 per loop body.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* io, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[8];
    uint64_t w[8];
    for (int i = 0; i < 8; i++) { a[i] = io[tid] + i * 0x9e3779b9u; w[i] = ((uint64_t)a[i] << 32) | (a[i] ^ 0x5555u); }
    const uint32_t b = io[tid + 1] | 1u, c = io[tid + 2];
    for (int it = 0; it < iters; it++) {
        if (OP == 0) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            BODY64(X)
#undef X
        } else if (OP == 1) {
#define X(i) asm volatile("v_and_b32 %0, 0x3fffffff, %0" : "+v"(a[i]));
            BODY64(X)
#undef X
        } else if (OP == 2) {
#define X(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
            BODY64(X)
#undef X
        } else if (OP == 3) {
#define X(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w[i]));
            BODY64(X)
#undef X
        } else if (OP == 4) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            BODY64(X)
#undef X
        } else if (OP == 5) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
            BODY64(X)
#undef X
        } else if (OP == 6) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            BODY64(X)
#undef X
        } else if (OP == 7) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            BODY64(X)
#undef X
        } else if (OP == 8) {      // one dependent chain of multiply-adds (as inside a field product)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[0]) : "v"(b), "v"(c) : "vcc");
            BODY64(X)
#undef X
        } else if (OP == 9) {
#define X(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            BODY64(X)
#undef X
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    io[tid] = r;
}

template <int OP>
static int run(const char* name, uint32_t* buf, int CUS) {
    const int iters = 2000;
    printf("%-22s", name);
    for (int wv = 1; wv <= 8; wv *= 2) {
        const int blocks = CUS * wv;          // 256-thread blocks: 4 waves each, one per SIMD -> wv waves per SIMD
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), 0, 0, buf, iters);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), 0, 0, buf, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave instructions per SIMD = wv * iters * 64; cycles (at a nominal 2.4 GHz) per wave instruction per SIMD
        printf("  w%d: %6.2f cyc", wv, best * 1e-3 * 2.4e9 / ((double)wv * iters * 64));
    }
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    uint32_t* buf;
    CK(hipMalloc(&buf, (size_t)CUS * 8 * 256 * 4 + 64));
    CK(hipMemset(buf, 0x5a, (size_t)CUS * 8 * 256 * 4 + 64));
    printf("cycles per wave instruction per SIMD (nominal 2.4 GHz), by waves per SIMD; 8 independent chains unless noted\n");
    run<0>("v_add_u32", buf, CUS);
    run<9>("v_sub_u32", buf, CUS);
    run<1>("v_and_b32 (literal)", buf, CUS);
    run<2>("v_lshrrev_b32", buf, CUS);
    run<6>("v_add3_u32", buf, CUS);
    run<7>("v_mov_b32", buf, CUS);
    run<3>("v_lshrrev_b64", buf, CUS);
    run<4>("v_mul_lo_u32", buf, CUS);
    run<5>("v_mad_u64_u32", buf, CUS);
    run<8>("v_mad_u64_u32 (1 chain)", buf, CUS);
    return 0;
}
