// Phase timing of k_small_msm (msm_small.hip.h) with wall_clock64 stamps per block: where the ~100 us of a small MSM go.
// Synthetic operands (random residues: the group-law formulas execute the same instructions as on curve points).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPORLA_SMALL_STAMPS -Iporla_amd/csrc tools/small_stamps.hip -o tools/small_stamps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "msm_small.hip.h"
using namespace porla;

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 3200;
    const int bits = argc > 2 ? atoi(argv[2]) : 256;
    const int c = argc > 3 ? atoi(argv[3]) : 0;
    using C = Bn254G1;
    using M = C::Fp;
    std::vector<uint8_t> sc(32 * (size_t)n, 0), pt(64 * (size_t)n);
    srand(1);
    for (size_t i = 0; i < sc.size(); i++) sc[i] = (i % 32) >= (size_t)(32 - bits / 8) ? rand() : 0;
    for (size_t i = 0; i < pt.size(); i++) pt[i] = (i % 32) == 0 ? (rand() & 0x1f) : rand();
    uint8_t *d_sc, *d_pt, *d_part;
    unsigned long long* d_st;
    uint32_t* h;
    hipMalloc(&d_sc, sc.size()); hipMalloc(&d_pt, pt.size());
    const size_t part_bytes = (size_t)SMALL_BLOCKS * SMALL_MAX_C * sizeof(XYZZ<M>);
    hipMalloc(&d_part, part_bytes + 1024); hipMemset(d_part, 0, part_bytes + 1024);
    hipMalloc(&d_st, SMALL_BLOCKS * 16 * 8);
    hipHostMalloc(&h, 64 * 1024, hipHostMallocMapped);
    hipMemcpy(d_sc, sc.data(), sc.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_pt, pt.data(), pt.size(), hipMemcpyHostToDevice);
    void* hd; hipHostGetDevicePointer(&hd, h, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(d_st, 0, SMALL_BLOCKS * 16 * 8);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_small_msm<C>), dim3(SMALL_BLOCKS), dim3(SMALL_THREADS), 0, 0, d_sc, d_pt, n, c, (XYZZ<M>*)d_part,
                           (uint32_t*)(d_part + part_bytes), (uint32_t*)hd, (XYZZ<M>*)((uint8_t*)hd + SMALL_HDR_WORDS * 4), (uint32_t)(rep + 1), (const uint8_t*)nullptr, 0u, d_st);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> st(SMALL_BLOCKS * 16);
        hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
        if (rep < 4) continue;
        printf("n=%u bits=%d: W=%u c=%u glv=%u  kernel %.1f us (events)\n", n, bits, h[1], h[2], h[3], ms * 1e3);
        unsigned long long t0 = ~0ull, tend = 0;
        for (int b = 0; b < SMALL_BLOCKS; b++) if (st[b * 16]) { if (st[b * 16] < t0) t0 = st[b * 16]; for (int k = 0; k < 10; k++) if (st[b * 16 + k] > tend) tend = st[b * 16 + k]; }
        printf("first stamp -> last stamp: %.1f us\n", (tend - t0) / 100.0);
        const char* names[] = {"or-scan", "digits", "sort", "accumulate", "fold T", "tree", "write sums", "arrive", "fold slices"};
        // the block with the longest lifetime and the finisher of window 0
        double sum[9] = {0}, mx[9] = {0}; int cnt = 0;
        for (int b = 0; b < SMALL_BLOCKS; b++) {
            if (!st[b * 16 + 7]) continue;
            cnt++;
            for (int k = 0; k < 7; k++) { double d = (st[b * 16 + k + 1] - st[b * 16 + k]) / 100.0; sum[k] += d; if (d > mx[k]) mx[k] = d; }
            if (st[b * 16 + 9]) { double d = (st[b * 16 + 9] - st[b * 16 + 8]) / 100.0; sum[8] += d; if (d > mx[8]) mx[8] = d; }
        }
        for (int k = 0; k < 9; k++) printf("  %-12s avg %7.2f us  max %7.2f us\n", names[k], sum[k] / (k == 8 ? (double)h[1] : cnt), mx[k]);
        printf("  blocks with work: %d, block start spread: ", cnt);
        unsigned long long smax = 0; for (int b = 0; b < SMALL_BLOCKS; b++) if (st[b * 16] > smax) smax = st[b * 16];
        printf("%.1f us\n", (smax - t0) / 100.0);
    }
    return 0;
}
