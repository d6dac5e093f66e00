"""GPU box: the known answers the REFERENCE ITSELF holds for its secp256k1 path, run straight through the HIP entry points -- no
oracle between the vector and the kernel.  Nothing here imports, links or executes anything under oracle/.

  test_ecmult_constants   porla/Utils/secp256k1_lib/tests.c:4715-4757  SHA-256 over the serialised x*G of 32 842 keys
                          == e4711b4d...b7b4 (tests.c:4729-4736); tests.c:4694-4695 computes each x*G with
                          secp256k1_ecmult_multi_var on ONE pair (cb -> (x, G)): here (i) every key as a one-pair
                          porla_secp256k1_msm_device call and (ii) all keys as one porla_fixed_base_commit batch over [G]
  run_ecmult_chain        tests.c:3493-3555  X <- xn*X + gn*G 20 000 times, each step a TWO-pair GPU MSM over (X, G);
                          the point after step 19 999 == D6E96687...2A88 (tests.c:3537-3544), and the closing identity
                          ae*A + ge*G == X (tests.c:3549-3554) as one more two-pair MSM
  test_ecmult_multi       tests.c:3816-4053  no points; 1- and 2-point products (with and without a G term); all-infinity points,
                          all-zero scalars, cancelling scalars on one point, one scalar on cancelling points, a scalar sum that
                          cancels; constant scalar / constant point; zero scalars among live ones; the exhaustive
                          s0*(t0*P) + s1*(t1*P) grid for 8 x 8 x 8 x 8 small signed values
  test_ecmult_multi_random  tests.c:4055-4221  300 random instances: 0 .. 128 inputs, dead terms (0*P, a*INF), a G term or not, half of
                          them built to sum to infinity by one compensating term

The reference's "random group elements" are k*G for seeded k here, so every expected value is (closed-form scalar) * G, computed on the
fixed-base [G] kernel that the first test ties to the reference's hash.  A G term (inp_g_sc) is one more (g_sc, G) pair: the engine's
entry point takes pairs only (Porla always passes g_sc = 0, Client.hpp:395,778; Server.hpp:842,848)."""
import hashlib
import random

import pytest

from tests import common  # noqa: F401  (ROOT on sys.path; nothing of the oracle is used in this file)

pytestmark = pytest.mark.gpu

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
P = 2**256 - 2**32 - 977
G = bytes.fromhex("79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798"      # group_impl.h:28-33
                  "483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8")
INF = bytes(64)


@pytest.fixture(scope="module")
def mx():
    from porla_amd import multiexp
    return multiexp


@pytest.fixture(scope="module")
def fb_g(mx):
    fb = mx.FixedBase("secp256k1", G, 1)
    yield fb
    fb.close()


def b32(k):
    return (k % N).to_bytes(32, "big")


def mul_g(fb, ks):
    """[k*G] on the fixed-base kernel: rows of ONE coefficient against the one-point base [G]"""
    ks = list(ks)
    out = fb.commit_host(b"".join(b32(k) for k in ks), len(ks), 1)
    return [out[64 * i:64 * i + 64] for i in range(len(ks))]


def neg(pt):
    return pt if pt == INF else pt[:32] + (P - int.from_bytes(pt[32:], "big")).to_bytes(32, "big")


def msm(mx, scalars, points):
    """ecmult_multi(szero, cb -> (scalars[i], points[i]), n) on the engine, host buffers as the include shim hands them over"""
    n = len(scalars)
    return mx.msm_host("secp256k1", b"".join(b32(s) for s in scalars), b"".join(points), n)


def constants_keys():
    keys = []
    for i in range(37):                       # tests.c:4738-4743
        keys += [i, (-i) % N]
    for i in range(256):                      # tests.c:4744-4751
        for j in range(1, 256, 2):
            keys.append((j << i) % N)
    return keys


def accumulate(points):
    acc = hashlib.sha256()
    for p in points:                          # tests.c:4702-4711: infinity as one zero byte, else the 65-byte uncompressed form
        acc.update(b"\x00" if p == INF else b"\x04" + p)
    return acc.hexdigest()


EXPECTED_CONSTANTS_HASH = "e4711b4d141e6848b7af472b4cd204143a7587601af96360d0cb1faa859ab7b4"      # tests.c:4729-4736


def test_ecmult_constants_as_one_fixed_base_batch(fb_g):
    keys = constants_keys()
    assert len(keys) == 74 + 256 * 128
    assert accumulate(mul_g(fb_g, keys)) == EXPECTED_CONSTANTS_HASH


def test_ecmult_constants_as_one_pair_msm_calls(mx):
    """tests.c:4694-4695: secp256k1_ecmult_multi_var(NULL, scratch, &rj5, &zero, cb -> (x, G), 1) for every key"""
    import torch
    keys = constants_keys()
    d_sc = torch.frombuffer(bytearray(b"".join(b32(k) for k in keys)), dtype=torch.uint8).cuda()
    d_g = torch.frombuffer(bytearray(G), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    pts = [mx.msm_device("secp256k1", d_sc.data_ptr() + 32 * i, d_g.data_ptr(), 1, stream) for i in range(len(keys))]
    assert pts[0] == INF and pts[1] == INF and pts[2] == G          # keys 0, -0, 1
    assert accumulate(pts) == EXPECTED_CONSTANTS_HASH


def test_ecmult_chain_through_two_pair_msms(mx):
    a = bytes.fromhex("8b30bbe9ae2a990696b22f670709dff3727fd8bc04d3362c6c7bf458e2846004"
                      "a357ae915c4a65281309edf20504740f0eb3343990216b4f81063cb65f2f7e0f")      # tests.c:3495-3500
    xn = 0x84cc5452f7fde1edb4d38a8ce9b1b84ccef31f146e569be9705d357a42985407
    gn = 0xa1e58d22553dcd42b23980625d4c57a96e9323d42b3152e5ca2c3990edc7c9de
    xf, gf = 0x1337, 0x7113
    ae, ge = 1, 0
    x = a
    for _ in range(20000):
        x = msm(mx, [xn, gn], [x, G])             # secp256k1_ecmult(&x, &x, &xn, &gn)
        ae = ae * xn % N
        ge = (ge * xn + gn) % N
        xn = xn * xf % N
        gn = gn * gf % N
    assert x.hex().upper() == ("D6E96687F9B10D092A6F35439D86CEBEA4535D0D409F53586440BD74B933E830"
                               "B95CBCA2C77DA786539BE8FD53354D2D3B4F566AE658045407ED6015EE1B2A88")      # tests.c:3537-3544
    assert msm(mx, [ae, ge], [a, G]) == x         # tests.c:3549-3554


def test_ecmult_multi_cases(mx, fb_g):
    rnd = random.Random(0x3816)
    rs = lambda: rnd.randrange(1, N)              # random_scalar_order

    def rp(count=1):
        """`count` random group elements with their discrete logs"""
        ks = [rs() for _ in range(count)]
        return ks, mul_g(fb_g, ks)

    # no points to multiply (tests.c:3830-3831): r = infinity
    assert mx.msm_host("secp256k1", b"", b"", 0) == INF

    # 1- and 2-point multiplies against ecmult (tests.c:3833-3874)
    for _ in range(16):
        s0, s1 = rs(), rs()
        (k,), (ptg,) = rp()
        assert msm(mx, [s0], [G]) == mul_g(fb_g, [s0])[0]                          # only G scalar
        assert msm(mx, [s0], [ptg]) == mul_g(fb_g, [s0 * k])[0]                    # 1-point
        want = mul_g(fb_g, [s0 * k + s1])[0]
        assert msm(mx, [s0, s1], [ptg, G]) == want                                 # 2-point
        assert msm(mx, [s1, s0], [G, ptg]) == want                                 # 2-point with G scalar: the G term as a pair

    # infinite outputs of various forms (tests.c:3876-3940)
    for _ in range(8):
        for size in (2, 10, 32):
            assert msm(mx, [rs() for _ in range(size)], [INF] * size) == INF       # infinity points
            _, pts = rp(size)
            assert msm(mx, [0] * size, pts) == INF                                 # zero scalars
        for size in (2, 10, 32):
            _, (ptg,) = rp()
            sc, pt = [], []
            for _i in range(16):
                s = rs()
                sc += [s, -s]
                pt += [ptg, ptg]
            assert msm(mx, sc[:size], pt[:size]) == INF                            # s*P + (-s)*P
            s = rs()
            _, pts = rp(16)
            pt = []
            for q in pts:
                pt += [q, neg(q)]
            assert msm(mx, [s] * size, pt[:size]) == INF                           # s*P + s*(-P)
        _, (ptg,) = rp()
        sc = [rs() for _ in range(31)]
        assert msm(mx, [sum(sc)] + [-s for s in sc], [ptg] * 32) == INF            # the scalars sum to zero

    # random points, constant scalar (tests.c:3942-3961); random scalars, constant point (tests.c:3963-3985)
    for _ in range(8):
        s = rs()
        ks, pts = rp(20)
        assert msm(mx, [s] * 20, pts) == mul_g(fb_g, [s * sum(ks)])[0]
        (k,), (ptg,) = rp()
        sc = [rs() for _ in range(20)]
        assert msm(mx, sc, [ptg] * 20) == mul_g(fb_g, [sum(sc) * k])[0]

    # zero scalars among live ones (tests.c:3987-4000)
    sc = [rs() for _ in range(20)]
    ks, pts = rp(20)
    sc[0] = 0
    assert msm(mx, sc, pts) == mul_g(fb_g, [sum(s * k for s, k in zip(sc, ks))])[0]
    sc[1] = sc[2] = sc[3] = sc[4] = 0
    assert msm(mx, sc[:6], pts[:6]) == mul_g(fb_g, [sc[5] * ks[5]])[0]
    assert msm(mx, sc[:5], pts[:5]) == INF

    # s0*(t0*P) + s1*(t1*P) exhaustively for small signed s0, s1, t0, t1 (tests.c:4002-4052)
    small = lambda i: ((i + 1) // 2) * (-1 if i & 1 else 1)       # 0, -1, 1, -2, 2, -3, 3, -4
    (k,), _ = rp()
    tp = mul_g(fb_g, [small(t) * k for t in range(8)])            # t*P for the eight values of t
    cases = [(t0, t1, s0, s1) for t0 in range(8) for t1 in range(8) for s0 in range(8) for s1 in range(8)]
    want = mul_g(fb_g, [(small(t0) * small(s0) + small(t1) * small(s1)) * k for t0, t1, s0, s1 in cases])
    for (t0, t1, s0, s1), w in zip(cases, want):
        assert msm(mx, [small(s0), small(s1)], [tp[t0], tp[t1]]) == w, (t0, t1, s0, s1)


def test_ecmult_multi_random_instances(mx, fb_g):
    """in the spirit of test_ecmult_multi_random (tests.c:4055-4221): few or many inputs (0 .. 128, exponentially distributed), few
    or many 0*P and a*INF terms, with or without a G term, an expected result that is infinity for about half of the instances --
    every point is k*G for a known k, so the expected result is (sum of the live s_i k_i) * G on the fixed-base [G] kernel, and a
    result forced to infinity by one compensating term (a r, -(1/a) G) as the reference builds it (tests.c:4110-4122)"""
    rnd = random.Random(0x4055)
    rs = lambda: rnd.randrange(1, N)
    cases = []
    for _ in range(300):
        num = rnd.randrange((1 << (2 + rnd.randrange(6))) + 1)          # 0 .. 4..128
        num_nonzero = rnd.randrange(num + 1)
        nonzero_result = rnd.getrandbits(1)
        g_nonzero = nonzero_result if num_nonzero == 0 else (1 if num_nonzero == 1 and not nonzero_result else rnd.getrandbits(1))
        terms = []                                                          # (scalar, discrete log of the point or None = infinity)
        live = num_nonzero
        if g_nonzero:
            terms.append((rs(), 1))                                         # the G term as one more pair
        while live > (0 if nonzero_result else 1):
            terms.append((rs(), rs()))
            live -= 1
        total = sum(s * k for s, k in terms) % N
        if not nonzero_result:
            if terms or live:
                # one compensating term brings the sum to infinity: scalar a, point -(total / a) G
                a = rs()
                terms.append((a, (-total * pow(a, -1, N)) % N))
                total = 0
        # the dead terms: 0 * P and a * INF, shuffled in between
        for _d in range(num - num_nonzero):
            terms.append((0, rs()) if rnd.getrandbits(1) else (rs(), None))
        rnd.shuffle(terms)
        cases.append((terms, total))
    # all the points and expected results in two batches of the fixed-base kernel
    logs = [k for terms, _ in cases for _, k in terms if k is not None]
    pts = iter(mul_g(fb_g, logs))
    want = mul_g(fb_g, [t for _, t in cases])
    zeros = 0
    for (terms, total), w in zip(cases, want):
        points = [INF if k is None else next(pts) for _, k in terms]
        got = msm(mx, [s for s, _ in terms], points) if terms else mx.msm_host("secp256k1", b"", b"", 0)
        assert got == w, (len(terms), total)
        zeros += w == INF
    assert 60 < zeros < 240                                                  # about half of the instances sum to infinity
