// Device-vs-host check and throughput of the reduced-radix field product (porla_amd/csrc/fe30.hip.h) against the portable
// Montgomery product of fe.hip.h: f30_mul(x, y) = x*y / 2^270, fe_mul_generic(x, y) = x*y / 2^256, so the two agree after one
// factor 2^14.  Operands: random reduced values, unreduced values up to 2^258 - 1, all-ones limbs, squares.
// Built by porla_amd/csrc/Makefile as porla_amd/fe30_check; run by tests/test_fe_gpu.py on the GPU box.  --bench: G products/s.
#include "host_curve.hpp"
#include "icc.hip.h"
#include "ec30.hip.h"
#include "inv30.hip.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace porla;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// operands travel as 9 raw 30-bit limbs (so that unreduced values with a wide top limb can be fed in)
template <class M>
__global__ void k_mul30(const uint32_t* a, const uint32_t* b, uint32_t* out, int n, int square) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F30<M> x, y;
    for (int k = 0; k < 9; k++) { x.v[k] = a[i * 9 + k]; y.v[k] = b[i * 9 + k]; }
    F30<M> z, zp;                               // the assembly must equal the portable form limb for limb
    if (square == 2) {                          // x y + c d with one reduction / one fold; (c, d) = the next lane's operands
        const int j = (i + 1) % n;
        F30<M> c, d;
        for (int k = 0; k < 9; k++) { c.v[k] = a[j * 9 + k]; d.v[k] = b[j * 9 + k]; }
        z = f30_mul2<M>(x, y, c, d);
        if constexpr (M::PSEUDO_MERSENNE) zp = f30_mul2_pm_portable<M>(x, y, c, d);
        else zp = f30_mul2_portable<M>(x, y, c, d);
        for (int k = 0; k < 9; k++) out[i * 9 + k] = z.v[k] | (z.v[k] != zp.v[k] ? 0x80000000u : 0u);
        return;
    }
    z = square ? f30_sqr<M>(x) : f30_mul<M>(x, y);
    if constexpr (M::PSEUDO_MERSENNE) zp = square ? f30_sqr_pm_portable<M>(x) : f30_mul_pm_portable<M>(x, y);
    else zp = square ? f30_sqr_portable<M>(x) : f30_mul_portable<M>(x, y);
    for (int k = 0; k < 9; k++) out[i * 9 + k] = z.v[k] | (z.v[k] != zp.v[k] ? 0x80000000u : 0u);
}
template <class M>
__global__ void k_roundtrip(const uint32_t* w, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t t[8], u[8];
    for (int k = 0; k < 8; k++) t[k] = w[i * 8 + k];
    F30<M> x = f30_unpack<M>(t);
    f30_pack<M>(u, x);
    for (int k = 0; k < 8; k++) out[i * 8 + k] = u[k];
}
template <class M, int SQ>
__global__ void k_bench(uint32_t* io, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    F30<M> x, y;
    for (int i = 0; i < 9; i++) { x.v[i] = io[tid * 9 + i] & F30_MASK; y.v[i] = (io[tid * 9 + i] ^ 0x01010101u) & F30_MASK; }
    x.v[8] &= 0xffff; y.v[8] &= 0xffff;
    for (int i = 0; i < iters; i++) {
        if (SQ) { x = f30_sqr<M>(x); y = f30_sqr<M>(y); }
        else { x = f30_mul<M>(x, y); y = f30_mul<M>(y, x); }
    }
    for (int i = 0; i < 9; i++) io[tid * 9 + i] = x.v[i] ^ y.v[i];
}

template <class M>
__global__ void k_madd30(uint32_t* io, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fe<M> ax, ay;
    for (int i = 0; i < 8; i++) { ax.v[i] = io[tid * 9 + i]; ay.v[i] = io[tid * 9 + i] ^ 0x01010101u; }
    ax.v[7] &= 0x0fffffff; ay.v[7] &= 0x0fffffff;
    XYZZ30<M> p; p.inf = true;
    bool flip = false;
    for (int i = 0; i < iters; i++) { xyzz30_madd_flip<M>(p, flip, f30_from_fe<M>(ax), f30_from_fe<M>(ay)); ax.v[1] += 1; }
    xyzz30_flip_finish<M>(p, flip);
    XYZZ<M> o = xyzz30_to_xyzz<M>(p);
    for (int i = 0; i < 8; i++) io[tid * 9 + i] = o.x.v[i] ^ o.y.v[i] ^ o.zz.v[i] ^ o.zzz.v[i];
}
template <class M>
__global__ void k_madd32(uint32_t* io, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Affine<M> a;
    for (int i = 0; i < 8; i++) { a.x.v[i] = io[tid * 9 + i]; a.y.v[i] = io[tid * 9 + i] ^ 0x01010101u; }
    a.x.v[7] &= 0x0fffffff; a.y.v[7] &= 0x0fffffff;
    XYZZ<M> p = xyzz_inf<M>();
    for (int i = 0; i < iters; i++) { xyzz_madd<M>(p, a); a.x.v[1] += 1; }
    for (int i = 0; i < 8; i++) io[tid * 9 + i] = p.x.v[i] ^ p.y.v[i] ^ p.zz.v[i] ^ p.zzz.v[i];
}

// host big-number helpers on 9 x 30-bit limbs
static void limbs_to_words(const uint32_t l[9], uint32_t w[9]) {   // 270 bits -> 9 words (w[8] = bits 256..269)
    for (int q = 0; q < 9; q++) w[q] = 0;
    for (int i = 0; i < 9; i++) {
        int o = 30 * i, q = o / 32, s = o % 32;
        uint64_t x = (uint64_t)l[i] << s;
        w[q] += 0;  // (limbs < 2^30 here: plain OR is enough)
        w[q] |= (uint32_t)x;
        if (q + 1 < 9) w[q + 1] |= (uint32_t)(x >> 32);
    }
}
// value (up to 2^288) mod p, binary long division
template <class M>
static Fe<M> mod_p(const uint32_t* w, int nwords) {
    uint32_t r[9] = {0};
    for (int bit = 32 * nwords - 1; bit >= 0; bit--) {
        for (int k = 8; k > 0; k--) r[k] = (r[k] << 1) | (r[k - 1] >> 31);
        r[0] = (r[0] << 1) | ((w[bit >> 5] >> (bit & 31)) & 1u);
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
// value of 9 limbs that may exceed 30 bits in the top limb (up to 2^258): accumulate into 10 words
template <class M>
static Fe<M> limbs_mod_p(const uint32_t l[9]) {
    uint32_t w[10] = {0};
    for (int i = 0; i < 9; i++) {
        int o = 30 * i, q = o / 32, s = o % 32;
        uint64_t x = (uint64_t)l[i] << s;
        uint64_t c = (uint64_t)w[q] + (uint32_t)x;
        w[q] = (uint32_t)c;
        c = (uint64_t)w[q + 1] + (uint32_t)(x >> 32) + (c >> 32);
        w[q + 1] = (uint32_t)c;
        for (int k = q + 2; k < 10 && (c >> 32); k++) { c = (uint64_t)w[k] + (c >> 32); w[k] = (uint32_t)c; }
    }
    return mod_p<M>(w, 10);
}

template <class M>
static int check(const char* name) {
    const int n = 1 << 16;
    std::vector<uint32_t> a(n * 9), b(n * 9), got(n * 9);
    std::mt19937_64 rng(11);
    for (int i = 0; i < n; i++) {
        int mode = i & 7;
        for (int k = 0; k < 9; k++) {
            a[i * 9 + k] = (uint32_t)rng() & F30_MASK;
            b[i * 9 + k] = (uint32_t)rng() & F30_MASK;
        }
        a[i * 9 + 8] &= 0x3fff; b[i * 9 + 8] &= 0x3fff;                                   // < 2^254
        if (mode == 1) { a[i * 9 + 8] = (uint32_t)rng() & 0x3ffff; b[i * 9 + 8] = (uint32_t)rng() & 0x3ffff; }   // unreduced, < 2^258
        if (mode == 2) for (int k = 0; k < 9; k++) { a[i * 9 + k] = k < 8 ? F30_MASK : 0x3ffff; b[i * 9 + k] = a[i * 9 + k]; }  // all ones
        if (mode == 3) for (int k = 0; k < 9; k++) a[i * 9 + k] = k < 8 ? F30_MASK : 0x3ffff;
        if (mode == 4) for (int k = 0; k < 9; k++) b[i * 9 + k] = a[i * 9 + k];
        if (mode == 5) for (int k = 0; k < 9; k++) a[i * 9 + k] = P30<M>::limb(k);         // p itself
        if (mode == 6) for (int k = 0; k < 9; k++) a[i * 9 + k] = (k == 0);                // one
        if (i == 7) for (int k = 0; k < 9; k++) a[i * 9 + k] = 0;
    }
    uint32_t *da, *db, *dout;
    CK(hipMalloc(&da, n * 36)); CK(hipMalloc(&db, n * 36)); CK(hipMalloc(&dout, n * 36));
    CK(hipMemcpy(da, a.data(), n * 36, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), n * 36, hipMemcpyHostToDevice));
    // 2^14 in Montgomery form (radix 2^256): fe_mul_generic(x, c14) = x * 2^14
    Fe<M> two14 = fe_zero<M>(); two14.v[0] = 1u << 14;
    Fe<M> r2; for (int k = 0; k < 8; k++) r2.v[k] = M::R2[k];
    Fe<M> c14 = fe_mul_generic<M>(two14, r2);
    int bad = 0;
    for (int square = 0; square < 3; square++) {          // 2: the two-product form f30_mul2 (x y + c d, one reduction)
        hipLaunchKernelGGL(k_mul30<M>, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n, square);
        CK(hipMemcpy(got.data(), dout, n * 36, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
            Fe<M> x = limbs_mod_p<M>(&a[i * 9]);
            Fe<M> y = square == 1 ? x : limbs_mod_p<M>(&b[i * 9]);
            Fe<M> want = fe_mul_generic<M>(x, y);                      // x*y / 2^256
            if (square == 2) {
                const int j = (i + 1) % n;
                want = fe_add<M>(want, fe_mul_generic<M>(limbs_mod_p<M>(&a[j * 9]), limbs_mod_p<M>(&b[j * 9])));
            }
            const uint32_t* z = &got[i * 9];
            bool limbs_ok = true;
            for (int k = 0; k < 9; k++) if (z[k] > F30_MASK) limbs_ok = false;
            Fe<M> zv = limbs_mod_p<M>(z);
            Fe<M> have = fe_mul_generic<M>(zv, c14);                   // (x*y / 2^270) * 2^14
            // value bound: z < p + 2^247 (operands < 2^258: a*b / 2^270 < 2^246, m*p / 2^270 < p)
            uint32_t zw[9]; limbs_to_words(z, zw);
            uint32_t lim[9]; uint64_t cy = 0;
            // (the two-product form: < p + 2^248)
            for (int k = 0; k < 9; k++) { cy += (uint64_t)(k < 8 ? M::P[k] : 0u) + (k == 7 ? (1u << (square == 2 ? 24 : 23)) : 0u); lim[k] = (uint32_t)cy; cy >>= 32; }
            bool le_p = false;
            for (int k = 8; k >= 0; k--) { if (zw[k] < lim[k]) { le_p = true; break; } if (zw[k] > lim[k]) break; }
            if (!limbs_ok || !le_p || !fe_eq<M>(want, have)) {
                if (bad < 5) printf("%s: mismatch at %d (square=%d mode=%d) limbs_ok=%d bound_ok=%d value_ok=%d\n", name, i, square, i & 7, limbs_ok, le_p, (int)fe_eq<M>(want, have));
                bad++;
            }
        }
    }
    // pack / unpack round trip
    std::vector<uint32_t> w(n * 8), w2(n * 8);
    for (auto& x : w) x = (uint32_t)rng();
    uint32_t *dw, *dw2;
    CK(hipMalloc(&dw, n * 32)); CK(hipMalloc(&dw2, n * 32));
    CK(hipMemcpy(dw, w.data(), n * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_roundtrip<M>, dim3(n / 256), dim3(256), 0, 0, dw, dw2, n);
    CK(hipMemcpy(w2.data(), dw2, n * 32, hipMemcpyDeviceToHost));
    if (memcmp(w.data(), w2.data(), n * 32)) { printf("%s: pack/unpack round trip differs\n", name); bad++; }
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout)); CK(hipFree(dw)); CK(hipFree(dw2));
    printf("%s: %d products + %d squares checked, %d mismatches\n", name, n, n, bad);
    return bad;
}

// special-form modulus: plain residues, f30_mul(x, y) = x*y mod p (value < 2^256 + 2^73, limbs normal)
template <class M>
static int check_pm(const char* name) {
    const int n = 1 << 16;
    std::vector<uint32_t> a(n * 9), b(n * 9), got(n * 9);
    std::mt19937_64 rng(13);
    for (int i = 0; i < n; i++) {
        int mode = i & 7;
        for (int k = 0; k < 9; k++) { a[i * 9 + k] = (uint32_t)rng() & F30_MASK; b[i * 9 + k] = (uint32_t)rng() & F30_MASK; }
        a[i * 9 + 8] &= 0xffff; b[i * 9 + 8] &= 0xffff;                                     // < 2^256
        if (mode == 1) { a[i * 9 + 8] = (uint32_t)rng() & 0x7ffff; b[i * 9 + 8] = (uint32_t)rng() & 0x7ffff; }   // unreduced, < 2^259
        if (mode == 2) for (int k = 0; k < 9; k++) { a[i * 9 + k] = k < 8 ? F30_MASK : 0x7ffff; b[i * 9 + k] = a[i * 9 + k]; }
        if (mode == 3) for (int k = 0; k < 9; k++) a[i * 9 + k] = k < 8 ? F30_MASK : 0x7ffff;
        if (mode == 4) for (int k = 0; k < 9; k++) b[i * 9 + k] = a[i * 9 + k];
        if (mode == 5) for (int k = 0; k < 9; k++) a[i * 9 + k] = P30<M>::limb(k);         // p itself
        if (mode == 6) for (int k = 0; k < 9; k++) a[i * 9 + k] = (k == 0);                // one
        if (i == 7) for (int k = 0; k < 9; k++) a[i * 9 + k] = 0;
    }
    uint32_t *da, *db, *dout;
    CK(hipMalloc(&da, n * 36)); CK(hipMalloc(&db, n * 36)); CK(hipMalloc(&dout, n * 36));
    CK(hipMemcpy(da, a.data(), n * 36, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), n * 36, hipMemcpyHostToDevice));
    int bad = 0;
    for (int square = 0; square < 3; square++) {          // 2: the two-product form f30_mul2 (x y + c d, one fold)
        hipLaunchKernelGGL(k_mul30<M>, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n, square);
        CK(hipMemcpy(got.data(), dout, n * 36, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
            auto full_product = [](const Fe<M>& x, const Fe<M>& y) {   // schoolbook product, then long division
                uint32_t t[16] = {0};
                for (int p = 0; p < 8; p++) {
                    uint64_t c = 0;
                    for (int q = 0; q < 8; q++) { c += (uint64_t)x.v[p] * y.v[q] + t[p + q]; t[p + q] = (uint32_t)c; c >>= 32; }
                    t[p + 8] = (uint32_t)c;
                }
                return mod_p<M>(t, 16);
            };
            Fe<M> x = limbs_mod_p<M>(&a[i * 9]);
            Fe<M> y = square == 1 ? x : limbs_mod_p<M>(&b[i * 9]);
            Fe<M> want = full_product(x, y);
            if (square == 2) {
                const int j = (i + 1) % n;
                want = fe_add<M>(want, full_product(limbs_mod_p<M>(&a[j * 9]), limbs_mod_p<M>(&b[j * 9])));
            }
            const uint32_t* z = &got[i * 9];
            bool limbs_ok = true;
            for (int k = 0; k < 8; k++) if (z[k] > F30_MASK) limbs_ok = false;
            if (z[8] > (1u << 16)) limbs_ok = false;                   // value < 2^256 + 2^73
            Fe<M> have = limbs_mod_p<M>(z);
            if (!limbs_ok || !fe_eq<M>(want, have)) {
                if (bad < 5) printf("%s: mismatch at %d (square=%d mode=%d) limbs_ok=%d value_ok=%d\n", name, i, square, i & 7, limbs_ok, (int)fe_eq<M>(want, have));
                bad++;
            }
        }
    }
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
    printf("%s: %d products + %d squares checked, %d mismatches\n", name, n, n, bad);
    return bad;
}

template <class F>
static double time_ms(F f, int reps = 3) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

// fe_inv_safegcd: a * a^-1 must be the unit of the Fe form (R mod p; 1 for the special-form modulus) for edge and random residues
template <class M>
__global__ void k_inv_check(const uint32_t* a8, uint32_t* out8, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<M> a;
    for (int k = 0; k < 8; k++) a.v[k] = a8[i * 8 + k];
    const Fe<M> r = fe_inv_safegcd<M>(a);
    const Fe<M> one = fe_mul_call<M>(a, r);
    for (int k = 0; k < 8; k++) out8[i * 8 + k] = one.v[k];
}
template <class M>
static int check_inv(const char* name) {
    const int N = 4096;
    std::vector<uint32_t> h((size_t)N * 8), o((size_t)N * 8);
    std::mt19937_64 rng(20261004);
    auto put = [&](int i, const uint32_t w[8]) { for (int k = 0; k < 8; k++) h[(size_t)i * 8 + k] = w[k]; };
    for (int i = 0; i < N; i++) {
        uint32_t w[9];
        for (int k = 0; k < 8; k++) w[k] = (uint32_t)rng();
        w[8] = 0;
        Fe<M> v = mod_p<M>(w, 9);
        bool zero = true;
        for (int k = 0; k < 8; k++) zero = zero && v.v[k] == 0;
        if (zero) v.v[0] = 1;
        put(i, v.v);
    }
    // edge residues: 1, 2, 3, p - 1, p - 2, (p - 1) / 2, (p + 1) / 2, 2^k, 2^k - 1, all-ones low limbs
    int e = 0;
    uint32_t w[8];
    auto small = [&](uint32_t x) { for (int k = 0; k < 8; k++) w[k] = 0; w[0] = x; put(e++, w); };
    small(1); small(2); small(3); small(0xffffffffu);
    for (uint32_t sub = 1; sub <= 2; sub++) { for (int k = 0; k < 8; k++) w[k] = M::P[k]; w[0] -= sub; put(e++, w); }   // P[0] >= 2 for both
    { uint32_t c = 0; for (int k = 7; k >= 0; k--) { uint32_t x = M::P[k]; w[k] = (x >> 1) | (c << 31); c = x & 1u; } put(e++, w);   // (p - 1) / 2
      w[0] += 1; put(e++, w); }                                                                                                   // (p + 1) / 2 (no carry: p = 3 mod 4 here)
    for (int bit : {1, 29, 30, 31, 32, 59, 60, 61, 119, 120, 239, 240, 241, 252}) {
        for (int k = 0; k < 8; k++) w[k] = 0;
        w[bit >> 5] = 1u << (bit & 31); put(e++, w);
        for (int k = 0; k < 8; k++) w[k] = k < (bit >> 5) ? 0xffffffffu : 0; w[bit >> 5] = (1u << (bit & 31)) - 1u; put(e++, w);
    }
    uint32_t *da, *dout;
    CK(hipMalloc(&da, h.size() * 4)); CK(hipMalloc(&dout, o.size() * 4));
    CK(hipMemcpy(da, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_inv_check<M>), dim3((N + 63) / 64), dim3(64), 0, 0, da, dout, N);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; i++)
        for (int k = 0; k < 8; k++) if (o[(size_t)i * 8 + k] != M::R1[k]) { bad++; break; }
    printf("%s: fe_inv_safegcd over %d residues (%d edge): %d wrong inverses\n", name, N, e, bad);
    (void)hipFree(da); (void)hipFree(dout);
    return bad;
}

// --peak: ONE JSON line with the back-to-back product / square rates (G per second, whole chip) of both base fields at 4
// waves per SIMD -- the occupancy the accumulation kernels run at -- and nothing else: bench.py prices `int_multiplier` with it
// in the same run, on the same box (boxes of the pool differ by a few per cent)
static int peak_line() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("{\"error\": \"no device\"}\n"); return 1; }
    const int CUS = prop.multiProcessorCount, it = 512;
    uint32_t* buf;
    if (hipMalloc(&buf, (size_t)CUS * 8 * 256 * 9 * 4) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
    std::vector<uint32_t> h((size_t)CUS * 8 * 256 * 9);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345);
    (void)hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // 4 waves per SIMD is what the accumulation kernels run at (103 .. 128 registers); 8 is the most the chip holds: the rate at
    // 8 is the multiplier's own ceiling, the rate at 4 what a kernel of that register footprint can reach
    // warm-up: ~0.1 s of the same work first -- measured cold, the first configuration came out 13 % low (159 instead of 183 G/s at
    // 4 waves per SIMD) while the clocks were still rising
    for (int r = 0; r < 40; r++) hipLaunchKernelGGL((k_bench<Bn254Fp, 0>), dim3(CUS * 8), dim3(256), 0, 0, buf, it);
    (void)hipDeviceSynchronize();
    printf("{\"compute_units\": %d", CUS);
    for (int w = 4; w <= 8; w *= 2) {
        const int blocks = CUS * w;
        auto rate = [&](double ms) { return (double)blocks * 256 * it * 2 / ms / 1e6; };
        const double bm = rate(time_ms([&] { hipLaunchKernelGGL((k_bench<Bn254Fp, 0>), dim3(blocks), dim3(256), 0, 0, buf, it); }, 5));
        const double bs = rate(time_ms([&] { hipLaunchKernelGGL((k_bench<Bn254Fp, 1>), dim3(blocks), dim3(256), 0, 0, buf, it); }, 5));
        const double sm = rate(time_ms([&] { hipLaunchKernelGGL((k_bench<Secp256k1Fp, 0>), dim3(blocks), dim3(256), 0, 0, buf, it); }, 5));
        const double ss = rate(time_ms([&] { hipLaunchKernelGGL((k_bench<Secp256k1Fp, 1>), dim3(blocks), dim3(256), 0, 0, buf, it); }, 5));
        printf(", \"waves_%d\": {\"bn254\": {\"mul_G_s\": %.2f, \"sqr_G_s\": %.2f}, \"secp256k1\": {\"mul_G_s\": %.2f, \"sqr_G_s\": %.2f}}",
               w, bm, bs, sm, ss);
    }
    printf("}\n");
    (void)hipFree(buf);
    return 0;
}

// --occupancy: the 8M + 2S product mix of a mixed addition (G products per second, whole chip) by waves per SIMD, 1 .. 8 -- what one
// more resident wave would buy a kernel that is bound by the multiplier
static int occupancy_lines() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int CUS = prop.multiProcessorCount, it = 512;
    uint32_t* buf;
    if (hipMalloc(&buf, (size_t)CUS * 8 * 256 * 9 * 4) != hipSuccess) return 1;
    std::vector<uint32_t> h((size_t)CUS * 8 * 256 * 9);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345);
    (void)hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 8; w++) {
        const int blocks = CUS * w;
        auto rate = [&](double ms) { return (double)blocks * 256 * it * 2 / ms / 1e6; };
        const double bm = rate(time_ms([&] { hipLaunchKernelGGL((k_bench<Bn254Fp, 0>), dim3(blocks), dim3(256), 0, 0, buf, it); }, 5));
        const double bs = rate(time_ms([&] { hipLaunchKernelGGL((k_bench<Bn254Fp, 1>), dim3(blocks), dim3(256), 0, 0, buf, it); }, 5));
        printf("waves/SIMD=%d  f30_mul<Bn254Fp> %.1f G/s  f30_sqr %.1f G/s  8M+2S mix %.1f G/s\n", w, bm, bs, 10.0 / (8.0 / bm + 2.0 / bs));
    }
    (void)hipFree(buf);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--peak")) return peak_line();
    if (argc > 1 && !strcmp(argv[1], "--occupancy")) return occupancy_lines();
    int bad = 0;
    bad += check_inv<Bn254Fp>("bn254_p");
    bad += check_inv<Secp256k1Fp>("secp256k1_p");
    bad += check<Bn254Fp>("bn254_p");
    bad += check<IccBn254Fr>("bn254_r");
    bad += check<IccFp>("p_icc");
    bad += check_pm<Secp256k1Fp>("secp256k1_p");
    if (argc > 1 && !strcmp(argv[1], "--bench")) {
        hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
        const int CUS = prop.multiProcessorCount;
        uint32_t* buf; (void)hipMalloc(&buf, (size_t)CUS * 2048 * 9 * 4);
        std::vector<uint32_t> h((size_t)CUS * 2048 * 9);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345);
        (void)hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int w = 1; w <= 8; w *= 2) {
            int blocks = CUS * w;
            const int it = 512;
            double ms = time_ms([&] { hipLaunchKernelGGL((k_bench<Bn254Fp, 0>), dim3(blocks), dim3(256), 0, 0, buf, it); });
            printf("f30_mul<Bn254Fp>  waves/SIMD=%d  %.3f ms  %.2f Gmul/s  (%.0f cycles/mul/wave @2.4GHz)\n", w, ms,
                   (double)blocks * 256 * it * 2 / ms / 1e6, ms * 1e-3 * 2.4e9 / (it * 2) / w);
            ms = time_ms([&] { hipLaunchKernelGGL((k_bench<Bn254Fp, 1>), dim3(blocks), dim3(256), 0, 0, buf, it); });
            printf("f30_sqr<Bn254Fp>  waves/SIMD=%d  %.3f ms  %.2f Gsqr/s\n", w, ms, (double)blocks * 256 * it * 2 / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL((k_bench<Secp256k1Fp, 0>), dim3(blocks), dim3(256), 0, 0, buf, it); });
            printf("f30_mul<Secp256k1> waves/SIMD=%d  %.3f ms  %.2f Gmul/s\n", w, ms, (double)blocks * 256 * it * 2 / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL((k_bench<Secp256k1Fp, 1>), dim3(blocks), dim3(256), 0, 0, buf, it); });
            printf("f30_sqr<Secp256k1> waves/SIMD=%d  %.3f ms  %.2f Gsqr/s\n", w, ms, (double)blocks * 256 * it * 2 / ms / 1e6);
            if (w <= 4) {
                ms = time_ms([&] { hipLaunchKernelGGL((k_madd30<Bn254Fp>), dim3(blocks), dim3(256), 0, 0, buf, 128); });
                printf("xyzz30_madd       waves/SIMD=%d  %.3f ms  %.2f Gadd/s\n", w, ms, (double)blocks * 256 * 128 / ms / 1e6);
                if (w == 4) {   // the same loop 8x longer: does the rate hold for the duration of a real accumulation kernel?
                    ms = time_ms([&] { hipLaunchKernelGGL((k_madd30<Bn254Fp>), dim3(blocks), dim3(256), 0, 0, buf, 1024); });
                    printf("xyzz30_madd (long) waves/SIMD=%d  %.3f ms  %.2f Gadd/s\n", w, ms, (double)blocks * 256 * 1024 / ms / 1e6);
                }
                ms = time_ms([&] { hipLaunchKernelGGL((k_madd32<Bn254Fp>), dim3(blocks), dim3(256), 0, 0, buf, 128); });
                printf("xyzz_madd (8x32)  waves/SIMD=%d  %.3f ms  %.2f Gadd/s\n", w, ms, (double)blocks * 256 * 128 / ms / 1e6);
            }
        }
    }
    return bad ? 1 : 0;
}
