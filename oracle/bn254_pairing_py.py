"""ORACLE (test infrastructure only) -- Python big-int restatement of the G2 / pairing half of the KZG plug-in:
SRS.G2 = (G2gen, tau * G2gen) (kzg.NewSRS, porla/main.go:46,66), their compressed wire form inside the SRS blob
(SRS.WriteTo / ReadFrom, main.go:48,67; blob size 32 n + 132, porla/Client/Client.hpp:350-357) and the pairing check behind
verify_proof (kzg.Verify, main.go:177-193).

Only tests/ and tests/golden/gen_*.py may import this file; the product path never does.

gnark-crypto v0.6.0 is not under /root/reference and cannot be fetched: the arithmetic follows the PUBLIC definition of
alt_bn128 (EIP-197): Fp2 = Fp[i]/(i^2 + 1), twist E'(Fp2): y^2 = x^3 + 3/(9 + i), G2 generator as in EIP-197, optimal ate
pairing with loop count 6x + 2 = 29793968203157093288 and the two Frobenius corrections, final exponent (p^12 - 1)/r.
It is deliberately a DIFFERENT formulation from porla_amd/csrc/pairing_host.hpp (which uses a 2-3-2 tower, projective line
steps, sparse products and an easy/hard final exponentiation): here Fp12 is the flat quotient Fp[w]/(w^12 - 18 w^6 + 82),
G2 points are mapped into E(Fp12) through the untwist (x, y) -> (x w^2, y w^3) and every line function is the generic affine
one, evaluated in Fp12 -- slow (about a second per pairing) and with nothing to get subtly wrong.
PARITY PIN: external known answers only (the reference holds none for this path): the EIP-197 generator, r * G2gen = infinity,
and go-ethereum's bn256Pairing precompile vector committed in tests/golden/eip197_kat.json.

Compressed G2 (gnark-crypto ecc/bn254/marshal.go, restated from its published format): 64 bytes X.A1 || X.A0 big-endian, top two
bits of byte 0 = 10 (Y is the lexicographically smallest of the two roots) / 11 (largest) / 01 (infinity); an Fp2 element is
"largest" when its A1 is > (p - 1)/2, or when A1 = 0 and A0 is > (p - 1)/2 (E2.LexicographicallyLargest).
"""
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
ATE_LOOP_COUNT = 29793968203157093288      # 6x + 2, x = 4965661367192848881
LOG_ATE_LOOP_COUNT = 63

# ---------------------------------------------------------------- Fp2 = Fp[i]/(i^2 + 1), elements (a0, a1) = a0 + a1 i
F2_ZERO, F2_ONE = (0, 0), (1, 0)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


def f2_sqrt(a):
    """a square root of a in Fp2 (p = 3 mod 4; the complex method), or None"""
    if a == F2_ZERO:
        return F2_ZERO
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(f2_mul(a1, a1), a)
    x0 = f2_mul(a1, a)
    if alpha == ((-1) % P, 0):
        r = f2_mul((0, 1), x0)
    else:
        r = f2_mul(f2_pow(f2_add(F2_ONE, alpha), (P - 1) // 2), x0)
    return r if f2_mul(r, r) == (a[0] % P, a[1] % P) else None


def f2_lex_largest(a):
    """E2.LexicographicallyLargest of gnark-crypto: strictly larger than its negation, A1 first"""
    if a[1] % P == 0:
        return a[0] % P > (P - 1) // 2
    return a[1] % P > (P - 1) // 2


# ---------------------------------------------------------------- G2 = E'(Fp2)[r], affine, None = infinity
B2 = f2_mul((3, 0), f2_inv((9, 1)))          # 3 / (9 + i)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,        # x = a0 + a1 i (EIP-197)
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def g2_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def g2_double(pt):
    if pt is None or pt[1] == F2_ZERO:
        return None
    x, y = pt
    xx = f2_mul(x, x)
    lam = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(y, y)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x), x)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x, x3)), y))


def g2_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    if p1[0] == p2[0]:
        return g2_double(p1) if p1[1] == p2[1] else None
    lam = f2_mul(f2_sub(p2[1], p1[1]), f2_inv(f2_sub(p2[0], p1[0])))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), p1[0]), p2[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(p1[0], x3)), p1[1]))


def g2_neg(pt):
    return None if pt is None else (pt[0], f2_neg(pt[1]))


def g2_mul(pt, k):
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, pt)
        pt = g2_double(pt)
        k >>= 1
    return acc


def g2_in_subgroup(pt):
    return g2_on_curve(pt) and g2_mul(pt, R) is None


def g2_compress(pt):
    """G2Affine.Bytes(): 64 bytes"""
    if pt is None:
        return bytes([0x40]) + bytes(63)
    b = bytearray(pt[0][1].to_bytes(32, "big") + pt[0][0].to_bytes(32, "big"))
    b[0] |= 0xC0 if f2_lex_largest(pt[1]) else 0x80
    return bytes(b)


def g2_decompress(buf):
    """G2Affine.SetBytes() of a compressed point; None = infinity; raises ValueError for a non-point"""
    buf = bytes(buf)
    flags = buf[0] & 0xC0
    if flags == 0x40:
        return None
    if flags == 0x00:
        raise ValueError("not a compressed point")
    x = (int.from_bytes(buf[32:64], "big") % P, int.from_bytes(bytes([buf[0] & 0x3F]) + buf[1:32], "big") % P)
    y = f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), B2))
    if y is None:
        raise ValueError("x is not on the twist")
    if f2_lex_largest(y) != (flags == 0xC0):
        y = f2_neg(y)
    return (x, y)


def g2_to_eip197(pt):
    """128 bytes x_im || x_re || y_im || y_re (EIP-197; = X.A1 || X.A0 || Y.A1 || Y.A0); infinity = zeros"""
    if pt is None:
        return bytes(128)
    return b"".join(v.to_bytes(32, "big") for v in (pt[0][1], pt[0][0], pt[1][1], pt[1][0]))


def g2_from_eip197(buf):
    v = [int.from_bytes(bytes(buf[32 * i:32 * i + 32]), "big") for i in range(4)]
    if not any(v):
        return None
    return ((v[1], v[0]), (v[3], v[2]))


def srs_g2_blob(tau):
    """the last 128 bytes of SRS.WriteTo (main.go:48): G2[0] = generator, G2[1] = tau * generator (kzg.NewSRS), compressed"""
    return g2_compress(G2_GEN) + g2_compress(g2_mul(G2_GEN, tau % R))


# ---------------------------------------------------------------- Fp12 = Fp[w]/(w^12 - 18 w^6 + 82), 12 coefficients, low first
F12_ONE = (1,) + (0,) * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):              # w^12 = 18 w^6 - 82
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return tuple(v % P for v in t[:12])


def f12_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def f12_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def f12_scalar(c):
    return (c % P,) + (0,) * 11


def _poly_deg(p):
    d = len(p) - 1
    while d >= 0 and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """extended Euclid in Fp[w] against the modulus polynomial"""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % P, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low) > 0:
        # r = high / low (polynomial quotient)
        dl = _poly_deg(low)
        temp = list(high)
        r = [0] * 13
        inv_lead = pow(low[dl], P - 2, P)
        for i in range(_poly_deg(temp) - dl, -1, -1):
            q = temp[dl + i] * inv_lead % P
            r[i] = q
            if q:
                for c in range(dl + 1):
                    temp[c + i] = (temp[c + i] - q * low[c]) % P
        nm, new = list(hm), list(high)
        for i in range(13):
            if lm[i] or low[i]:
                for j in range(13 - i):
                    if r[j]:
                        nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                        new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    if low[0] == 0:
        raise ZeroDivisionError("Fp12 inverse of zero")
    inv0 = pow(low[0], P - 2, P)
    return tuple(v * inv0 % P for v in lm[:12])


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_mul(a, a)
        e >>= 1
    return r


_W2 = (0, 0, 1) + (0,) * 9
_W3 = (0, 0, 0, 1) + (0,) * 8


def untwist(pt):
    """E'(Fp2) -> E(Fp12): y^2 = x^3 + 3.  i = w^6 - 9 (so that (9 + i) = w^6), then (x, y) -> (x w^2, y w^3)"""
    if pt is None:
        return None
    (x0, x1), (y0, y1) = pt
    nx = ((x0 - 9 * x1) % P, 0, 0, 0, 0, 0, x1 % P, 0, 0, 0, 0, 0)
    ny = ((y0 - 9 * y1) % P, 0, 0, 0, 0, 0, y1 % P, 0, 0, 0, 0, 0)
    return (f12_mul(nx, _W2), f12_mul(ny, _W3))


def _e12_double(pt):
    x, y = pt
    lam = f12_mul(f12_mul(f12_scalar(3), f12_mul(x, x)), f12_inv(f12_add(y, y)))
    x3 = f12_sub(f12_sub(f12_mul(lam, lam), x), x)
    return (x3, f12_sub(f12_mul(lam, f12_sub(x, x3)), y))


def _e12_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    if p1[0] == p2[0]:
        return _e12_double(p1) if p1[1] == p2[1] else None
    lam = f12_mul(f12_sub(p2[1], p1[1]), f12_inv(f12_sub(p2[0], p1[0])))
    x3 = f12_sub(f12_sub(f12_mul(lam, lam), p1[0]), p2[0])
    return (x3, f12_sub(f12_mul(lam, f12_sub(p1[0], x3)), p1[1]))


def _linefunc(p1, p2, t):
    """the line through p1 and p2 (tangent when equal) evaluated at t, all in E(Fp12)"""
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    elif y1 == y2:
        m = f12_mul(f12_mul(f12_scalar(3), f12_mul(x1, x1)), f12_inv(f12_add(y1, y1)))
    else:
        return f12_sub(xt, x1)
    return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))


def miller_loop(q_g2, p_g1):
    """the optimal ate Miller function f_{6x+2,Q}(P) with its two Frobenius line corrections, BEFORE the final exponentiation"""
    if q_g2 is None or p_g1 is None:
        return F12_ONE
    q = untwist(q_g2)
    p = (f12_scalar(p_g1[0]), f12_scalar(p_g1[1]))
    r = q
    f = F12_ONE
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(r, r, p))
        r = _e12_double(r)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _linefunc(r, q, p))
            r = _e12_add(r, q)
    q1 = (f12_pow(q[0], P), f12_pow(q[1], P))
    nq2 = (f12_pow(q1[0], P), f12_sub(f12_scalar(0), f12_pow(q1[1], P)))
    f = f12_mul(f, _linefunc(r, q1, p))
    r = _e12_add(r, q1)
    f = f12_mul(f, _linefunc(r, nq2, p))
    return f


def final_exponentiate(f):
    return f12_pow(f, (P ** 12 - 1) // R)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for pairs [(G1 affine tuple or None, G2 affine or None)] -- the EIP-197 predicate and kzg.Verify's"""
    f = F12_ONE
    for p_g1, q_g2 in pairs:
        f = f12_mul(f, miller_loop(q_g2, p_g1))
    return final_exponentiate(f) == F12_ONE
