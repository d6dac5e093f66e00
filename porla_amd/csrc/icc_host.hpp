// Host helpers shared by the ICC data encode (icc.hip) and the MAC-side encode (mac_fft.hip).
#pragma once
#include "host_curve.hpp"
#include "icc.hip.h"

namespace porla {

inline int ilog2u(size_t n) { int l = 0; while (n >>= 1) l++; return l; }
inline uint64_t rev_bits(uint64_t x, int n) { uint64_t r = 0; for (int i = 0; i < n; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

// w = GENERATOR^((p_icc - 1)/(2N)) mod p_icc (Server.hpp:214-216), Montgomery form
inline Fe<IccFp> icc_root(size_t n) {
    Fe<IccFp> g;
    for (int i = 0; i < 8; i++) g.v[i] = IccGen::G[i];
    g = fe_to_mont<IccFp>(g);
    // (p-1)/(2N) = 207 * 2^(247 - log2 N)
    int sh = 247 - ilog2u(n);
    uint32_t e[8] = {0};
    uint64_t v = 207;
    int limb = sh >> 5, off = sh & 31;
    uint64_t lo = v << off;
    e[limb] = (uint32_t)lo;
    if (limb + 1 < 8) e[limb + 1] = (uint32_t)(lo >> 32);
    return h_fe_pow<IccFp>(g, e);
}


// wt = w^reverse_bits(write_step % N, height-1) mod p_icc (Server.hpp:1494), Montgomery form mod p_icc
inline Fe<IccFp> icc_wt(size_t n, unsigned long long write_step) {
    const int logn = ilog2u(n);
    const int height = logn + 1;
    uint64_t ex = rev_bits(write_step % n, height - 1);
    uint32_t e[8] = {(uint32_t)ex, (uint32_t)(ex >> 32), 0, 0, 0, 0, 0, 0};
    return h_fe_pow<IccFp>(icc_root(n), e);
}

}  // namespace porla
