#!/usr/bin/env python3
"""Derives and self-checks the GLV constants used by porla_amd/csrc/glv.hip.h, and prints them as C++ initialisers.

Curves y^2 = x^3 + b have the endomorphism phi(x, y) = (beta*x, y) = lambda*(x, y) with beta^3 = 1 mod p, lambda^3 = 1 mod n.
A scalar k is split as k = k1 + lambda*k2 (mod n) with |k1|, |k2| ~ sqrt(n) using a reduced basis (a1, b1), (a2, b2) of the
lattice {(a, b): a + lambda*b = 0 mod n}:   c1 = round(b2*k/n), c2 = round(-b1*k/n),
    k1 = k - c1*a1 - c2*a2,   k2 = -c1*b1 - c2*b2                (exact integers, no reduction mod n)
The roundings use g_i = round(2^S * |b_j| / n) and c_i = (k*g_i + 2^(S-1)) >> S, S = 384 or less so that g_i < 2^256 (the method of libsecp256k1's
scalar_split_lambda, porla/Utils/secp256k1_lib/scalar_impl.h:123-156; any rounding error only moves (k1, k2) by a lattice
vector, the identity k = k1 + lambda*k2 holds by construction).  secp256k1's lambda/beta are the reference's constants
(scalar_impl.h:64-67, group_impl.h:651-654); BN254's are derived here (gnark uses the same endomorphism internally).
"""
import random
import sys

CURVES = {
    "Bn254": dict(p=21888242871839275222246405745257275088696311157297823662689037894645226208583,
                  n=21888242871839275222246405745257275088548364400416034343698204186575808495617, b=3, G=(1, 2)),
    "Secp256k1": dict(p=2**256 - 2**32 - 977,
                      n=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141, b=7,
                      G=(0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
                         0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8),
                      lam=0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72,
                      beta=0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE),
}


def ec_add(p, a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % p == 0: return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, p) % p
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, p) % p
    x = (lam * lam - a[0] - b[0]) % p
    return (x, (lam * (a[0] - x) - a[1]) % p)


def ec_mul(p, a, k):
    acc = None
    for bit in bin(k)[2:]:
        acc = ec_add(p, acc, acc)
        if bit == "1": acc = ec_add(p, acc, a)
    return acc


def cube_roots(m):
    for g in range(2, 100):
        r = pow(g, (m - 1) // 3, m)
        if r != 1:
            return r, r * r % m
    raise RuntimeError


def derive(name, c):
    p, n, G = c["p"], c["n"], c["G"]
    if "lam" in c:
        lam, beta = c["lam"], c["beta"]
    else:
        lam = beta = None
        for l in cube_roots(n):
            for bt in cube_roots(p):
                if ec_mul(p, G, l) == (bt * G[0] % p, G[1]):
                    lam, beta = l, bt
        assert lam is not None
    assert (lam * lam + lam + 1) % n == 0 and (beta * beta + beta + 1) % p == 0
    assert ec_mul(p, G, lam) == (beta * G[0] % p, G[1])
    # reduced basis from the extended Euclidean algorithm on (n, lam)
    rs, ts = [n, lam], [0, 1]
    while rs[-1] != 0:
        q = rs[-2] // rs[-1]
        rs.append(rs[-2] - q * rs[-1]); ts.append(ts[-2] - q * ts[-1])
    l = max(i for i in range(len(rs)) if rs[i] * rs[i] >= n)
    a1, b1 = rs[l + 1], -ts[l + 1]
    cand = [(rs[l], -ts[l]), (rs[l + 2], -ts[l + 2])]
    a2, b2 = min(cand, key=lambda v: v[0] * v[0] + v[1] * v[1])
    for a, b in ((a1, b1), (a2, b2)):
        assert (a + lam * b) % n == 0
    det = a1 * b2 - a2 * b1
    assert abs(det) == n
    if det < 0:            # orient the basis so that det = +n
        a2, b2 = -a2, -b2
    shift = 384
    while ((max(abs(b1), abs(b2)) << shift) + n // 2) // n >= 1 << 256:   # g must fit 256 bits
        shift -= 1
    g1 = ((abs(b2) << shift) + n // 2) // n        # c1 = round(b2 k / n)
    g2 = ((abs(b1) << shift) + n // 2) // n        # c2 = round(-b1 k / n)
    return dict(name=name, p=p, n=n, lam=lam, beta=beta, a1=a1, b1=b1, a2=a2, b2=b2, g1=g1, g2=g2, shift=shift)


M256 = (1 << 256) - 1


def split_model(d, k):
    """bit-for-bit model of glv_split in glv.hip.h: unsigned 256-bit words, constants by magnitude + compile-time sign"""
    sh = d["shift"]
    assert d["g1"] < 1 << 256 and d["g2"] < 1 << 256
    c1 = (k * d["g1"] + (1 << (sh - 1))) >> sh       # approximates |b2| k / n
    c2 = (k * d["g2"] + (1 << (sh - 1))) >> sh       # approximates |b1| k / n
    assert c1 < 1 << 128 and c2 < 1 << 128
    s1 = 1 if d["b2"] >= 0 else -1                   # c1_true = s1 * c1
    s2 = 1 if -d["b1"] >= 0 else -1                  # c2_true = s2 * c2
    k1 = (k - s1 * c1 * d["a1"] - s2 * c2 * d["a2"]) & M256
    k2 = (-s1 * c1 * d["b1"] - s2 * c2 * d["b2"]) & M256
    out = []
    for v in (k1, k2):
        neg = v >> 255
        mag = ((-v) & M256) if neg else v
        out.append((mag, neg))
    return out


def check(d, trials=200000):
    n, lam = d["n"], d["lam"]
    rnd = random.Random(1)
    worst = 0
    ks = [0, 1, 2, n - 1, n - 2, lam, n - lam, (n - 1) // 2, (n + 1) // 2, 1 << 128, (1 << 128) - 1, 1 << 255 if (1 << 255) < n else n >> 1]
    ks += [rnd.randrange(n) for _ in range(trials)]
    ks += [(rnd.randrange(1 << 130) * d["a1"] + rnd.randrange(1 << 20)) % n for _ in range(2000)]
    for k in ks:
        (m1, n1), (m2, n2) = split_model(d, k)
        k1 = -m1 if n1 else m1
        k2 = -m2 if n2 else m2
        assert (k1 + lam * k2 - k) % n == 0, hex(k)
        worst = max(worst, m1, m2)
    return worst.bit_length()


def limbs(v, count=8):
    return "{" + ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xffffffff) for i in range(count)) + "}"


if __name__ == "__main__":
    for name, c in CURVES.items():
        d = derive(name, c)
        bits = check(d, 200000 if len(sys.argv) < 2 else int(sys.argv[1]))
        bound1 = (abs(d["a1"]) + abs(d["a2"])) // 2 + 2
        bound2 = (abs(d["b1"]) + abs(d["b2"])) // 2 + 2
        print("// %s: lambda = 0x%x" % (name, d["lam"]))
        print("//   beta = 0x%x" % d["beta"])
        print("//   basis (a1, b1) = (%d, %d), (a2, b2) = (%d, %d)" % (d["a1"], d["b1"], d["a2"], d["b2"]))
        print("//   proven bound (|a1|+|a2|)/2, (|b1|+|b2|)/2 (+ rounding): %d / %d bits; worst seen over the self-check: %d bits"
              % (bound1.bit_length(), bound2.bit_length(), bits))
        print("struct Glv%s {" % name)
        print("    static constexpr uint32_t BETA[8] = %s;   // plain" % limbs(d["beta"]))
        print("    static constexpr int SHIFT = %d;" % d["shift"])
        print("    static constexpr uint32_t G1[8] = %s;     // round(2^SHIFT |b2| / n)" % limbs(d["g1"]))
        print("    static constexpr uint32_t G2[8] = %s;     // round(2^SHIFT |b1| / n)" % limbs(d["g2"]))
        for nm in ("a1", "b1", "a2", "b2"):
            print("    static constexpr uint32_t %s[5] = %s;   static constexpr bool %s_NEG = %s;"
                  % (nm.upper(), limbs(abs(d[nm]), 5), nm.upper(), "true" if d[nm] < 0 else "false"))
        print("    static constexpr int BITS = %d;   // |k1|, |k2| < 2^BITS" % max(bound1.bit_length(), bound2.bit_length()))
        print("};")
