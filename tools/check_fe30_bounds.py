#!/usr/bin/env python3
"""Worst-case column sums of fe30.hip.h:f30_mul / f30_sqr for every modulus the engine uses: all must stay below 2^64.
Operand limbs 0..7 <= 2^30 - 1, limb 8 <= TOP - 1 (value < 2^258 -> TOP = 2^18); m_i <= 2^30 - 1; the modulus limbs are exact."""
MODULI = {
    "bn254_p": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "bn254_r": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "p_icc": 207 * 2**248 + 1,
}
# the ICC kernel of icc30.hip.h multiplies an UNREDUCED butterfly output (< 2^263: limb 8 < 2^23) by a twiddle (a product's result,
# < p + 2^248: limb 8 < 2^17) modulo p_icc, the BN254 group order and the secp256k1 group order
ICC_MODULI = {
    "p_icc": 207 * 2**248 + 1,
    "bn254_r": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "secp256k1_n": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
}
MASK = 2**30 - 1
def check(name, p, top_bits=18, top_bits_b=None):
    P = [(p >> (30 * i)) & MASK for i in range(9)]
    amax = [MASK] * 8 + [2**top_bits - 1]
    bmax = [MASK] * 8 + [2**(top_bits_b or top_bits) - 1]
    carry = 0
    worst = 0
    for k in range(17):
        t = carry
        for i in range(9):
            j = k - i
            if 0 <= j <= 8:
                t += amax[i] * bmax[j]
                if not (k < 9 and i == k):
                    pass
        # reduction products m_i * P_{k-i}, i <= min(k, 8); for k < 9 this includes m_k * P_0
        for i in range(9):
            j = k - i
            if 0 <= j <= 8:
                t += MASK * P[j]
        worst = max(worst, t)
        carry = t >> 30
    ok = worst < 2**64
    print("%-8s worst column %.4f x 2^64  %s" % (name, worst / 2**64, "ok" if ok else "OVERFLOW"))
    return ok
MUL2_CHAIN = (6, 8)     # = tools/gen_fe30_asm.py:MUL2_CHAIN = fe30.hip.h:F30_MUL2_CHAIN_LO / _HI
def check_mul2(name, p, top_bits=18, chain=MUL2_CHAIN):
    """f30_mul2: t = a b + reduction (+ c d outside the chained columns, + one 30-bit digit of u inside them, + u's carry after them);
    u = the products of c d in the chained columns with its own carries"""
    k0, k1 = chain
    P = [(p >> (30 * i)) & MASK for i in range(9)]
    amax = [MASK] * 8 + [2**top_bits - 1]
    ct = cu = 0
    worst_t = worst_u = 0
    for k in range(17):
        ab = sum(amax[i] * amax[k - i] for i in range(9) if 0 <= k - i <= 8)
        mp = sum(MASK * P[k - i] for i in range(9) if 0 <= k - i <= 8)
        if k0 <= k <= k1:
            u = cu + ab
            worst_u, cu = max(worst_u, u), u >> 30
            t = ct + MASK + ab + mp
        else:
            t = ct + 2 * ab + mp + (cu if k == k1 + 1 else 0)
        worst_t, ct = max(worst_t, t), t >> 30
    ok = worst_t < 2**64 and worst_u < 2**64
    print("%-8s mul2: worst t column %.4f x 2^64, worst u column %.4f x 2^64  %s" % (name, worst_t / 2**64, worst_u / 2**64, "ok" if ok else "OVERFLOW"))
    return ok
PM_MUL2_CHAIN = 7        # = tools/gen_fe30_asm.py:PM_MUL2_CHAIN = fe30.hip.h:F30_PM_MUL2_CHAIN
def check_pm_mul2(top_bits=19, chain=PM_MUL2_CHAIN):
    """f30_mul2_pm (special-form modulus, no reduction terms in the columns): both products' terms in t, except the chained column,
    whose c d terms run in u; then the fold of a double-width value < 2^519: R[9] < 2^26"""
    amax = [MASK] * 8 + [2**top_bits - 1]
    ct = cu = 0
    worst = 0
    for k in range(17):
        ab = sum(amax[i] * amax[k - i] for i in range(9) if 0 <= k - i <= 8)
        if k == chain:
            cu = ab >> 30
            t = ct + ab + MASK
        else:
            t = ct + 2 * ab + (cu if k == chain + 1 else 0)
        worst, ct = max(worst, t), t >> 30
    unchained = max(2 * sum(amax[i] * amax[k - i] for i in range(9) if 0 <= k - i <= 8) for k in range(17)) + 2**35
    # the fold: hi = bits 270.. of the sum (< 2^519 / 2^270), times 2^14 (2^32 + FOLD) with FOLD < 2^16
    hi_max = (2 * (2**259 - 1) ** 2) >> 270
    r9_ok = (hi_max * (2**14 * (2**32 + 2**16)) + 2**270) >> 270 < 2**26
    ok = worst < 2**64 and r9_ok
    print("special-form mul2: worst column %.6f x 2^64 (a single accumulator would reach %.10f), fold R[9] < 2^26: %s  %s"
          % (worst / 2**64, unchained / 2**64, r9_ok, "ok" if ok else "OVERFLOW"))
    return ok
if __name__ == "__main__":
    import sys
    ok = all([check(n, p) for n, p in MODULI.items()])
    ok = all([check_mul2(n, p) for n, p in MODULI.items() if n != "p_icc"]) and ok
    ok = check_pm_mul2() and ok
    ok = all([check("icc:" + n, p, 23, 17) for n, p in ICC_MODULI.items()]) and ok
    # result bound of the ICC product: a b / 2^270 + p < p + 2^249 for a < 2^263, b < 2^256: limb 8 stays far below 2^30
    sys.exit(0 if ok else 1)
