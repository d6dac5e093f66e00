"""CPU: pins the secp256k1 oracle (oracle/secp256k1_ref.c) against the reference's OWN known-answer tests,
restated here byte for byte:
  test_ecmult_constants  porla/Utils/secp256k1_lib/tests.c:4715-4757   expected hash tests.c:4729-4736
  run_ecmult_chain       porla/Utils/secp256k1_lib/tests.c:3493-3555   expected point tests.c:3537-3544
plus algebraic checks in the style of test_ecmult_multi (tests.c:3816-4053)."""
import ctypes
import hashlib

from tests import common

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def mul_g(ks):
    L = common.oracle()
    out = ctypes.create_string_buffer(64 * len(ks))
    L.oracle_secp256k1_mul_g_batch(b"".join(k.to_bytes(32, "big") for k in ks), ctypes.c_size_t(len(ks)), out, common.ncpu())
    return [out.raw[64 * i:64 * i + 64] for i in range(len(ks))]


def test_ecmult_constants_hash():
    """tests.c:4715-4757: keys i, -i for i in 0..36, then (j * 2^i) mod n for i < 256, odd j < 256"""
    keys = []
    for i in range(37):
        keys += [i, (-i) % N]
    for i in range(256):
        for j in range(1, 256, 2):
            keys.append((j << i) % N)
    acc = hashlib.sha256()
    for p in mul_g(keys):
        acc.update(b"\x00" if p == bytes(64) else b"\x04" + p)   # tests.c:4702-4711
    assert acc.hexdigest() == "e4711b4d141e6848b7af472b4cd204143a7587601af96360d0cb1faa859ab7b4"


def test_ecmult_chain_expected_point():
    """tests.c:3493-3555, state after iteration i == 19999 (20 000 updates)"""
    L = common.oracle()
    a = bytes.fromhex("8b30bbe9ae2a990696b22f670709dff3727fd8bc04d3362c6c7bf458e2846004"
                      "a357ae915c4a65281309edf20504740f0eb3343990216b4f81063cb65f2f7e0f")
    xn = bytes.fromhex("84cc5452f7fde1edb4d38a8ce9b1b84ccef31f146e569be9705d357a42985407")
    gn = bytes.fromhex("a1e58d22553dcd42b23980625d4c57a96e9323d42b3152e5ca2c3990edc7c9de")
    out = ctypes.create_string_buffer(64)
    L.oracle_secp256k1_ecmult_chain(a, xn, gn, 0x1337, 0x7113, 20000, out)
    assert out.raw.hex().upper() == ("D6E96687F9B10D092A6F35439D86CEBEA4535D0D409F53586440BD74B933E830"
                                     "B95CBCA2C77DA786539BE8FD53354D2D3B4F566AE658045407ED6015EE1B2A88")


def test_multi_algebra():
    """in the style of test_ecmult_multi (tests.c:3876-3940): zero scalars, infinity points, cancelling pairs,
    bucket method == naive sum, == single scalar product on collinear inputs"""
    L = common.oracle()
    n = 300
    ks = [int.from_bytes(hashlib.sha256(b"pt%d" % i).digest(), "big") % N for i in range(n)]
    pts = mul_g(ks)
    sc = [int.from_bytes(hashlib.sha256(b"sc%d" % i).digest(), "big") for i in range(n)]
    scb = b"".join(s.to_bytes(32, "big") for s in sc)
    out1, out2 = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    L.oracle_secp256k1_multi(scb, b"".join(pts), ctypes.c_size_t(n), out1, 1, 1)
    L.oracle_secp256k1_multi(scb, b"".join(pts), ctypes.c_size_t(n), out2, 3, 0)
    want = mul_g([sum(s * k for s, k in zip(sc, ks)) % N])[0]
    assert out1.raw == want and out2.raw == want
    # zero scalars / infinity points contribute nothing; P and -P cancel
    p = 2**256 - 2**32 - 977
    neg0 = pts[0][:32] + (p - int.from_bytes(pts[0][32:], "big")).to_bytes(32, "big")
    L.oracle_secp256k1_multi((5).to_bytes(32, "big") * 2 + bytes(32) + scb[:32], pts[0] + neg0 + pts[1] + bytes(64),
                             ctypes.c_size_t(4), out1, 1, 0)
    assert out1.raw == bytes(64)
    assert L.oracle_secp256k1_on_curve(want) == 1


def test_bench_style_vectors_match_the_committed_fixture():
    """tests/golden/secp256k1_golden.json: ecmult_multi on the reference bench's inputs at Porla's sizes, naive and bucket
    method of the C oracle, and the closed form of the bench teardown"""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "secp256k1_golden.json")))
    for case in gold["cases"]:
        n = case["n"]
        sc, pt = common.secp_bench_scalars(n), common.secp_bench_points(n)
        want = bytes.fromhex(case["result_xy"])
        assert common.oracle_secp_msm(sc, pt, n) == want
        assert common.oracle_secp_msm(sc, pt, n, naive=True) == want
        assert common.secp_bench_expected(sc, n) == want
        assert case["result_compressed"] == ("03" if want[63] & 1 else "02") + want[:32].hex()
