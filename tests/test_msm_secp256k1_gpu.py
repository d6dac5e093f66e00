"""Parity of the HIP secp256k1 MSM (the IPA scheme's secp256k1_ecmult_multi_var with g_sc = 0,
porla/Utils/secp256k1_lib/ecmult_impl.h:814-860) against the oracle, through the C ABI.  Inputs follow the
reference's own bench generator (bench_ecmult.c:233-247, 328-337); parity is on the normalised affine point."""
import hashlib

import pytest

from tests import common

pytestmark = pytest.mark.gpu

LAMBDA = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72  # the curve endomorphism's eigenvalue (tools/gen_glv.py); scalars around it stress the GLV split


@pytest.fixture(autouse=True, params=["glv", "plain", "general"])
def scalar_split(request):
    """every test runs with the GLV scalar split (default) and with plain full-width windows"""
    from porla_amd import lib
    lib.porla_gpu_set_msm_glv(0 if request.param == "plain" else 1)
    lib.porla_gpu_set_msm_small(0 if request.param == "general" else 1, 0)   # "general": the single-launch path off
    yield request.param
    lib.porla_gpu_set_msm_glv(-1)
    lib.porla_gpu_set_msm_small(1, 0)
P = 2**256 - 2**32 - 977


@pytest.fixture(scope="module")
def mx():
    from porla_amd import multiexp
    return multiexp


@pytest.fixture(scope="module")
def inputs():
    n = 1 << 15   # the reference bench's POINTS = 32768
    return common.secp_bench_scalars(n), common.secp_bench_points(n)


# sizes the reference really uses: 16 / 8 per commitment call, 176 and 1408 in audits
@pytest.mark.parametrize("n", [0, 1, 2, 8, 16, 128, 176, 1408, 1 << 15])
def test_bench_inputs_match_oracle_and_closed_form(mx, inputs, n):
    sc, pt = inputs
    got = mx.msm_host("secp256k1", sc[:32 * n], pt[:64 * n], n)
    assert got == common.secp_bench_expected(sc, n)
    if n <= 1408:
        assert got == common.oracle_secp_msm(sc, pt, n, naive=True)


@pytest.mark.parametrize("c", [2, 4, 7, 12, 15, 16, 17])
def test_every_window_width(mx, inputs, c):
    from porla_amd import lib
    sc, pt = inputs
    n = 1500
    lib.porla_gpu_set_msm_window(c)
    try:
        got = mx.msm_host("secp256k1", sc[:32 * n], pt[:64 * n], n)
    finally:
        lib.porla_gpu_set_msm_window(0)
    assert got == common.secp_bench_expected(sc, n)


def test_edge_cases(mx, inputs):
    """tests.c:3876-3940 style: zero scalars, infinity points, cancelling pairs, scalars >= n, n-1, repeated points"""
    sc, pt = inputs
    N = common.SECP_N
    p0 = pt[:64]
    neg0 = p0[:32] + (P - int.from_bytes(p0[32:], "big")).to_bytes(32, "big")
    vals = [0, 1, N - 1, N, N + 5, 2**256 - 1, 7, 7, 9, 9, LAMBDA, LAMBDA + 1, N - LAMBDA, LAMBDA * LAMBDA % N,
            (N - 1) // 2, (N + 1) // 2, (1 << 128) - 1, 1 << 128, 0x3086d221a7d46bcde86c90e49284eb15,
            0xe4437ed6010e88286f547fa90abfe4c3, 0x114ca50f7a8e2f3f657c1108d9d44cfd8]
    pts = [pt[64 * i:64 * i + 64] for i in range(6)] + [p0, neg0, p0, p0] + [pt[64 * i:64 * i + 64] for i in range(6, 17)]
    scs = b"".join(v.to_bytes(32, "big") for v in vals) + sc[:32]
    ptb = b"".join(pts) + bytes(64)
    n = len(vals) + 1
    assert mx.msm_host("secp256k1", scs, ptb, n) == common.oracle_secp_msm(scs, ptb, n, naive=True)
    assert mx.msm_host("secp256k1", (7).to_bytes(32, "big") * 2, p0 + neg0, 2) == bytes(64)


def test_audit_like(mx, inputs):
    """abs(int32) coefficients over repeated MAC points (Server.hpp:604-732, 842-848)"""
    import random
    sc, pt = inputs
    rnd = random.Random(3)
    n = 1408
    scs = b"".join(bytes(28) + rnd.getrandbits(31).to_bytes(4, "big") for _ in range(n))
    pts = b"".join(pt[64 * (i % 11):64 * (i % 11) + 64] for i in range(n))
    assert mx.msm_host("secp256k1", scs, pts, n) == common.oracle_secp_msm(scs, pts, n)


def test_device_pointer_and_partials(mx, inputs):
    import torch
    sc, pt = inputs
    n = 1 << 15
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    want = common.secp_bench_expected(sc, n)
    assert mx.msm_device("secp256k1", d_sc.data_ptr(), d_pt.data_ptr(), n, stream) == want
    parts = b""
    for g in range(8):
        lo, hi = n * g // 8, n * (g + 1) // 8
        parts += mx.msm_device("secp256k1", d_sc.data_ptr() + 32 * lo, d_pt.data_ptr() + 64 * lo, hi - lo, stream, partial=True)
    assert mx.jac_sum("secp256k1", parts, 8) == want


def test_full_size_2_20(mx):
    """BASELINE.json config 4: 2^20-point secp256k1 ecmult_multi; expected = (sum s_i 2^i) G"""
    import torch
    n = 1 << 20
    sc, pt = common.secp_bench_scalars(n), common.secp_bench_points(n)
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    got = mx.msm_device("secp256k1", d_sc.data_ptr(), d_pt.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert got == common.secp_bench_expected(sc, n)


def test_committed_bench_style_vectors(mx):
    """tests/golden/secp256k1_golden.json through the engine (host-buffer entry point)"""
    import json
    import os
    gold = json.load(open(os.path.join(common.ROOT, "tests", "golden", "secp256k1_golden.json")))
    for case in gold["cases"]:
        n = case["n"]
        sc, pt = common.secp_bench_scalars(n), common.secp_bench_points(n)
        assert mx.msm_host("secp256k1", sc, pt, n) == bytes.fromhex(case["result_xy"])
