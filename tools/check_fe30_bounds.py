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
SECP_P = 2**256 - 2**32 - 977


def check_value_ranges():
    """the operand ranges the group law of ec30.hip.h leans on (comments there: X <= 5p, Y <= 4p, U = 2Y <= 8p in a doubling,
    D = Q - X3 + 6p <= 7p + eps in an addition) against what a product accepts: value < 2^258 (limb 8 < 2^18) for BN254,
    < 2^259 (limb 8 < 2^19) for secp256k1; and what the lazy MEMORY form can hold (256 bits: BN254 stores X <= 5p, Y <= 4p
    unreduced, secp256k1 reduces on the way out); eps = the slack of a product's result (2^247 / 2^50)"""
    ok = True
    bn = MODULI["bn254_p"]
    for name, p, budget, eps in (("bn254_p", bn, 2**258, 2**247), ("secp256k1_p", SECP_P, 2**259, 2**50)):
        worst = max(2 * (4 * p), 7 * p + 8 * eps, 5 * p + 4 * eps)       # U = 2 Y (Y <= 4p), D, X
        good = worst < budget
        print("%-12s largest product operand of the group law %.3f x budget  %s" % (name, worst / budget, "ok" if good else "TOO LARGE"))
        ok = ok and good
    mem = 5 * bn + 2**247 < 2**256 and 4 * bn < 2**256
    print("bn254_p      lazy memory form: X <= 5p + eps and Y <= 4p fit 256 bits: %s" % ("ok" if mem else "NO"))
    return ok and mem


def check_sub_tables():
    """borrow-free subtractions a + (K p' - b) of fe30.hip.h: every limb of the K p table of f30_sub exceeds a normal limb of b
    (limbs 0..7) and the top limb exceeds b's for b <= (K - 1) p + 2^247; f30_sub_twice (a - 2 b, allowance doubled) stays
    inside 32 bits per limb, carries included"""
    ok = True
    for name, p in (("bn254_p", MODULI["bn254_p"]), ("secp256k1_p", SECP_P)):
        for K in (2, 3, 4, 5, 6):
            kp = K * p
            T = [(kp >> (30 * i)) & MASK for i in range(8)] + [kp >> 240]
            T[0] += 2**30
            for i in range(1, 8):
                T[i] += 2**30 - 1
            T[8] -= 1
            assert sum(t << (30 * i) for i, t in enumerate(T)) == kp
            b_top = ((K - 1) * p + 2**247) >> 240
            good = all(T[i] >= MASK for i in range(8)) and T[8] >= b_top and max(T[:8]) + MASK + 3 < 2**32
            ok = ok and good
        # f30_sub_twice<3>: X3 = MM - 2 S + 3 p of a doubling
        kp = 3 * p
        T = [(kp >> (30 * i)) & MASK for i in range(8)] + [kp >> 240]
        T2 = [T[0] + 2**31] + [t + 2**31 - 2 for t in T[1:8]] + [T[8] - 2]
        assert sum(t << (30 * i) for i, t in enumerate(T2)) == kp
        two_b_top = (2 * (p + 2**247)) >> 240
        good = all(T2[i] >= 2 * MASK for i in range(8)) and T2[8] >= two_b_top and max(T2[:8]) + MASK + 3 < 2**32          # + a normal limb of a + a carry of at most 3
        print("%-12s subtraction tables (K = 2..6, and the doubled one of f30_sub_twice<3>): %s" % (name, "ok" if ok and good else "BAD"))
        ok = ok and good
    return ok


if __name__ == "__main__":
    import sys
    ok = all([check(n, p) for n, p in MODULI.items()])
    ok = all([check_mul2(n, p) for n, p in MODULI.items() if n != "p_icc"]) and ok
    ok = check_pm_mul2() and ok
    ok = all([check("icc:" + n, p, 23, 17) for n, p in ICC_MODULI.items()]) and ok
    ok = check_value_ranges() and ok
    ok = check_sub_tables() and ok
    # result bound of the ICC product: a b / 2^270 + p < p + 2^249 for a < 2^263, b < 2^256: limb 8 stays far below 2^30
    sys.exit(0 if ok else 1)
