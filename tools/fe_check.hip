// Device-vs-host check of the gfx950 assembly field products (fe_mul_gfx950.inc) against portable references -- the C++
// Montgomery product fe_mul_generic, and for the special-form secp256k1 field a schoolbook product with binary long
// division -- for every field of the engine, on random, unreduced and all-ones-limb operands.
// Built by porla_amd/csrc/Makefile as porla_amd/fe_check; run by tests/test_fe_gpu.py on the GPU box.
#include "host_curve.hpp"
#include "icc.hip.h"
#include <cstdio>
#include <random>
#include <vector>
using namespace porla;

// independent reference for the special-form field (plain residues): schoolbook product, then binary long division by p
template <class M>
Fe<M> ref_mulmod(const Fe<M>& a, const Fe<M>& b) {
    uint32_t t[16] = {0};
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 8; j++) { c += (uint64_t)a.v[i] * b.v[j] + t[i + j]; t[i + j] = (uint32_t)c; c >>= 32; }
        t[i + 8] = (uint32_t)c;
    }
    uint32_t r[9] = {0};   // remainder, < 2p < 2^257
    for (int bit = 511; bit >= 0; bit--) {
        for (int k = 8; k > 0; k--) r[k] = (r[k] << 1) | (r[k - 1] >> 31);
        r[0] = (r[0] << 1) | ((t[bit >> 5] >> (bit & 31)) & 1u);
        uint32_t d[9];
        uint64_t br = 0;
        for (int k = 0; k < 9; k++) {
            uint64_t x = (uint64_t)r[k] - (k < 8 ? M::P[k] : 0u) - br;
            d[k] = (uint32_t)x;
            br = (x >> 63) & 1;
        }
        if (!br) for (int k = 0; k < 9; k++) r[k] = d[k];
    }
    Fe<M> o;
    for (int k = 0; k < 8; k++) o.v[k] = r[k];
    return o;
}
template <class M>
Fe<M> ref_mul(const Fe<M>& a, const Fe<M>& b) { return M::PSEUDO_MERSENNE ? ref_mulmod<M>(a, b) : fe_mul_generic<M>(a, b); }

template <class M>
__global__ void k_mul(const Fe<M>* a, const Fe<M>* b, Fe<M>* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_mul<M>(a[i], b[i]);
}

template <class M>
int check(const char* name) {
    const int n = 1 << 16;
    std::vector<Fe<M>> a(n), b(n), want(n), got(n);
    std::mt19937_64 rng(7);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 8; k++) { a[i].v[k] = (uint32_t)rng(); b[i].v[k] = (uint32_t)rng(); }
        int mode = i & 7;
        if (mode == 1) for (int k = 0; k < 8; k++) a[i].v[k] = 0xffffffffu;                       // unreduced all-ones
        if (mode == 2) for (int k = 0; k < 7; k++) { a[i].v[k] = 0xffffffffu; b[i].v[k] = 0xffffffffu; }
        if (mode == 3) { for (int k = 0; k < 8; k++) a[i].v[k] = 0xffffffffu; b[i] = a[i]; }
        if (mode == 4) { b[i] = a[i]; }
        if (mode >= 5) { fe_reduce_plain<M>(a[i].v, 8); fe_reduce_plain<M>(b[i].v, 8); }            // reduced operands
        if (mode == 6) for (int k = 0; k < 8; k++) b[i].v[k] = M::R2[k];                            // x * R2 with x reduced
        if (mode == 0) for (int k = 0; k < 8; k++) b[i].v[k] = M::R2[k];                            // x * R2 with x unreduced
        if (mode == 2 || mode == 3) fe_reduce_plain<M>(b[i].v, 8);                                   // one operand < p
        want[i] = ref_mul<M>(a[i], b[i]);
    }
    Fe<M>*da, *db, *dout;
    hipMalloc(&da, n * sizeof(Fe<M>)); hipMalloc(&db, n * sizeof(Fe<M>)); hipMalloc(&dout, n * sizeof(Fe<M>));
    hipMemcpy(da, a.data(), n * sizeof(Fe<M>), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * sizeof(Fe<M>), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_mul<M>), dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    hipMemcpy(got.data(), dout, n * sizeof(Fe<M>), hipMemcpyDeviceToHost);
    int bad = 0, bad_mode[8] = {0};
    for (int i = 0; i < n; i++) if (!fe_eq<M>(got[i], want[i])) {
        if (!bad) {
            auto pr = [](const char* t, const Fe<M>& f) { printf("  %s ", t); for (int k = 7; k >= 0; k--) printf("%08x", f.v[k]); printf("\n"); };
            printf("  first mismatch (operand mode %d):\n", i & 7);
            pr("a   ", a[i]); pr("b   ", b[i]); pr("want", want[i]); pr("got ", got[i]);
        }
        bad++; bad_mode[i & 7]++;
    }
    printf("%-16s %d mismatches of %d  (by operand mode:", name, bad, n);
    for (int m = 0; m < 8; m++) printf(" %d", bad_mode[m]);
    printf(")\n");
    hipFree(da); hipFree(db); hipFree(dout);
    return bad;
}

// operands that are partly COMPILE-TIME constants: the register allocator may then tie an asm input to an accumulator
// operand of equal value unless those are early-clobber (the bug this guards against: "+v" instead of "+&v")
template <class M>
__global__ void k_const_zero_limbs(const Fe<M>* a, const Fe<M>* b, Fe<M>* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<M> x;
    for (int k = 0; k < 8; k++) x.v[k] = k < 3 ? a[i].v[k] : 0;
    out[i] = fe_mul<M>(x, b[i]);
}
template <class M>
__global__ void k_const_r2(const Fe<M>* a, Fe<M>* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<M> r2;
    for (int k = 0; k < 8; k++) r2.v[k] = M::R2[k];
    out[i] = fe_from_mont<M>(fe_mul<M>(a[i], r2));
}

template <class M>
int check_const(const char* name) {
    const int n = 1 << 13;
    std::vector<Fe<M>> a(n), b(n), w1(n), w2(n), g(n);
    std::mt19937_64 rng(3);
    Fe<M> r2;
    for (int k = 0; k < 8; k++) r2.v[k] = M::R2[k];
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 8; k++) { a[i].v[k] = (uint32_t)rng(); b[i].v[k] = (uint32_t)rng(); }
        fe_reduce_plain<M>(b[i].v, 8);
        Fe<M> x;
        for (int k = 0; k < 8; k++) x.v[k] = k < 3 ? a[i].v[k] : 0;
        w1[i] = ref_mul<M>(x, b[i]);
        Fe<M> one = fe_zero<M>(); one.v[0] = 1;
        w2[i] = ref_mul<M>(ref_mul<M>(a[i], r2), one);
    }
    Fe<M>*da, *db, *dout;
    hipMalloc(&da, n * sizeof(Fe<M>)); hipMalloc(&db, n * sizeof(Fe<M>)); hipMalloc(&dout, n * sizeof(Fe<M>));
    hipMemcpy(da, a.data(), n * sizeof(Fe<M>), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * sizeof(Fe<M>), hipMemcpyHostToDevice);
    int bad = 0;
    hipLaunchKernelGGL((k_const_zero_limbs<M>), dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    hipMemcpy(g.data(), dout, n * sizeof(Fe<M>), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) bad += !fe_eq<M>(g[i], w1[i]);
    hipLaunchKernelGGL((k_const_r2<M>), dim3(n / 256), dim3(256), 0, 0, da, dout, n);
    hipMemcpy(g.data(), dout, n * sizeof(Fe<M>), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) bad += !fe_eq<M>(g[i], w2[i]);
    printf("%-16s constant operands: %d mismatches of %d\n", name, bad, 2 * n);
    hipFree(da); hipFree(db); hipFree(dout);
    return bad;
}

template <class M>
__global__ void k_wide(const uint32_t* a, Fe<M>* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_from_mont<M>(icc_reduce_wide<M, 19>(a + 19 * i));
}

template <class M>
int check_wide(const char* name) {
    const int n = 1 << 14;
    std::vector<uint32_t> a(19 * n);
    std::vector<Fe<M>> want(n), got(n);
    std::mt19937_64 rng(9);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 19; k++) a[19 * i + k] = (uint32_t)rng();
        if (i & 1) { a[19 * i + 18] = 0; a[19 * i + 17] &= 0xff; }
        if ((i & 3) == 2) for (int k = 0; k < 16; k++) a[19 * i + k] = 0xffffffffu;
        want[i] = fe_from_mont<M>(icc_reduce_wide<M, 19>(&a[19 * i]));
    }
    uint32_t* da; Fe<M>* dout;
    hipMalloc(&da, a.size() * 4); hipMalloc(&dout, n * sizeof(Fe<M>));
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_wide<M>), dim3(n / 256), dim3(256), 0, 0, da, dout, n);
    hipMemcpy(got.data(), dout, n * sizeof(Fe<M>), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) if (!fe_eq<M>(got[i], want[i])) bad++;
    printf("%-16s wide reduction: %d mismatches of %d\n", name, bad, n);
    hipFree(da); hipFree(dout);
    return bad;
}

int main() {
    int bad = 0;
    bad += check<Bn254Fp>("Bn254Fp");
    bad += check<Bn254Fr>("Bn254Fr");
    bad += check<Secp256k1Fp>("Secp256k1Fp");
    bad += check<IccFp>("IccFp");
    bad += check<IccBn254Fr>("IccBn254Fr");
    bad += check<IccSecp256k1Fn>("IccSecp256k1Fn");
    bad += check_const<Bn254Fp>("Bn254Fp");
    bad += check_const<Secp256k1Fp>("Secp256k1Fp");
    bad += check_const<IccFp>("IccFp");
    bad += check_const<IccBn254Fr>("IccBn254Fr");
    bad += check_wide<IccFp>("IccFp");
    bad += check_wide<IccBn254Fr>("IccBn254Fr");
    bad += check_wide<IccSecp256k1Fn>("IccSecp256k1Fn");
    printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 1 : 0;
}
