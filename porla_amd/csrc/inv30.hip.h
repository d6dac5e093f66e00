// Modular inversion on the device by division steps (used by the finish kernels: fixed_base.hip.h:k_fb_finish,
// mac_fft.hip.h:k_mac_finish / store_affine_be).  Checked directly by tools/fe30_check.hip (a * a^-1 over edge and random residues,
// both base fields) and modelled on Python integers, register ranges asserted, by tools/safegcd_model.py.
#pragma once
#include "fe.hip.h"
#include "fe30.hip.h"

namespace porla {

// Inversion without the exponentiation: Bernstein-Yang division steps ("Fast constant-time gcd computation and modular
// inversion", 2019) on (f, g) = (p, V), 30 steps at a time on the low words with the steps' 2 x 2 transition matrix applied to
// the full-size f, g and -- modulo p -- to the cofactors d, e (d V = f mod p throughout).  741 steps suffice for 256-bit inputs
// (Theorem 11.2: floor((49 * 256 + 57) / 17)); at most 25 rounds of 30 are run, every lane of a wave the same instructions.  Numbers are nine
// signed limbs of 30 bits (limbs 0..7 in [0, 2^30), limb 8 carries the sign); all products fit 64-bit signed accumulators
// because |u| + |v| <= 2^30 for a matrix row.  ~22 k simple instructions against the 381 dependent field products of a^(p-2)
// (fixed_base.hip.h:fe_inv_dev: 0.32 ms for a lone wave in the 8 x 32-bit form, 0.18 ms with the reduced-radix product).  tools/safegcd_model.py is the same procedure on Python integers with the 32 / 64-bit ranges asserted.
// f30_inv_safegcd_raw: a = any non-zero 256-bit value V; returns the integer V^-1 mod p plus a multiple of p, non-negative and
// below 4 p, in normal 30-bit limbs -- the caller's product with a constant puts it into the form it needs
// (fe_inv_safegcd below: the Fe form, constant Fp::INV_OUT_30).
template <class M>
__device__ __noinline__ F30<M> f30_inv_safegcd_raw(Fe<M> a) {
    constexpr int32_t M30 = (int32_t)F30_MASK;
    int32_t f[9], g[9], d[9], e[9];
    {
        const F30<M> gv = f30_unpack<M>(a.v);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            f[i] = (int32_t)P30<M>::limb(i);
            g[i] = (int32_t)gv.v[i];
            d[i] = 0;
            e[i] = i == 0 ? 1 : 0;
        }
    }
    int32_t delta = 1;
#pragma unroll 1
    for (int round = 0; round < 25; round++) {
        // once g = 0 a round changes nothing that matters: its matrix is [[2^30, 0], [0, 1]], so f stays and d stays modulo p (a
        // negative d becomes d + p) -- the wave leaves as soon as every lane is there (~20 rounds for random inputs, not 25); the
        // result bytes are those of the full 25 rounds
        if (round >= 8) {
            int32_t nz = g[0];
#pragma unroll
            for (int i = 1; i < 9; i++) nz |= g[i];
            if (!__any(nz != 0)) break;
        }
        uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << 30), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
        int32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll 6
        for (int i = 0; i < 30; i++) {
            const bool godd = g0 & 1u;
            const bool sw = godd && delta > 0;                   // (delta, f, g) -> (1 - delta, g, (g - f) / 2)
            const uint32_t nf = sw ? g0 : f0;
            uint32_t ng = sw ? 0u - f0 : g0;
            const int32_t nu = sw ? q : u, nv = sw ? r : v;
            int32_t nq = sw ? -u : q, nr = sw ? -v : r;
            delta = sw ? -delta : delta;
            ng += godd ? nf : 0u;                                // else (1 + delta, f, (g + (g mod 2) f) / 2)
            nq += godd ? nu : 0;
            nr += godd ? nv : 0;
            delta += 1;
            f0 = nf; g0 = ng >> 1;
            u = nu * 2; v = nv * 2; q = nq; r = nr;              // 2^(i+1) (f', g') = [[u, v], [q, r]] (f, g)
        }
        // (f, g) <- [[u, v], [q, r]] (f, g) / 2^30, exactly
        {
            int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0];
            int64_t cg = (int64_t)q * f[0] + (int64_t)r * g[0];
            cf >>= 30; cg >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                cf += (int64_t)u * f[i] + (int64_t)v * g[i];
                cg += (int64_t)q * f[i] + (int64_t)r * g[i];
                f[i - 1] = (int32_t)cf & M30; cf >>= 30;
                g[i - 1] = (int32_t)cg & M30; cg >>= 30;
            }
            f[8] = (int32_t)cf; g[8] = (int32_t)cg;
        }
        // (d, e) <- the same matrix times (d, e), divided by 2^30 modulo p: a negative d or e first counts as d + p (so the
        // combination lies in (-2^30 p, 2^30 p)), then k p with k in [0, 2^30) is taken off to clear the low 30 bits:
        // the results stay in (-2 p, p)
        {
            const int32_t p0 = (int32_t)P30<M>::limb(0);
            int32_t md = (d[8] < 0 ? u : 0) + (e[8] < 0 ? v : 0);
            int32_t me = (d[8] < 0 ? q : 0) + (e[8] < 0 ? r : 0);
            int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
            int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
            // k = x p0^-1 mod 2^30 for x = low bits of (c + m p0);  -p^-1 = P30::INV, so k = -(x INV)
            const uint32_t xd = ((uint32_t)cd + (uint32_t)md * (uint32_t)p0) & F30_MASK;
            const uint32_t xe = ((uint32_t)ce + (uint32_t)me * (uint32_t)p0) & F30_MASK;
            md -= (int32_t)((0u - xd * P30<M>::INV) & F30_MASK);
            me -= (int32_t)((0u - xe * P30<M>::INV) & F30_MASK);
            cd += (int64_t)md * p0; ce += (int64_t)me * p0;
            cd >>= 30; ce >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                const int32_t pi = (int32_t)P30<M>::limb(i);
                cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)md * pi;
                ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)me * pi;
                const int32_t dl = (int32_t)cd & M30, el = (int32_t)ce & M30;
                cd >>= 30; ce >>= 30;
                d[i - 1] = dl; e[i - 1] = el;
            }
            d[8] = (int32_t)cd; e[8] = (int32_t)ce;
        }
    }
    // g = 0, f = +-1, d V = f: the inverse is f d in (-2 p, 2 p); + 2 p makes it a non-negative number below 4 p, which the
    // reduced-radix product takes as it is
    const bool neg = f[8] < 0;
    F30<M> x;
    {
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int64_t t = (int64_t)(neg ? -d[i] : d[i]) + 2 * (int64_t)P30<M>::limb(i) + c;
            if (i < 8) { x.v[i] = (uint32_t)t & F30_MASK; c = t >> 30; }
            else x.v[i] = (uint32_t)t;
        }
    }
    return x;
}
// a: a non-zero residue in the Fe form; returns its inverse in the Fe form (see Fp::INV_OUT_30).
template <class M>
__device__ __forceinline__ Fe<M> fe_inv_safegcd(Fe<M> a) {
    return f30_to_fe_canonical<M>(f30_mul<M>(f30_inv_safegcd_raw<M>(a), f30_const<M>(M::INV_OUT_30)));
}

}  // namespace porla
