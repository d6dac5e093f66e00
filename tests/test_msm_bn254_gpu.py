"""Parity of the HIP BN254 G1 MSM against the oracle, through the C ABI (GPU box only).

Mirrors what the reference's callers do: bn254_multi_exp(result, points, scalars, n)
(porla/Utils/utils.h:277-292 -> compute_multi_exp, porla/main.go:118-138).  Bar: bit-exact 64-byte output.
"""
import ctypes
import os

import pytest

from tests import common

pytestmark = pytest.mark.gpu

LAMBDA = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23  # the curve endomorphism's eigenvalue (tools/gen_glv.py); scalars around it stress the GLV split


@pytest.fixture(autouse=True, params=["glv", "plain", "general"])
def scalar_split(request):
    """every test runs with the GLV scalar split and with plain full-width windows (inputs of up to 4096 pairs then take the
    single-launch path of msm_small.hip.h), and once more with that path switched off (the general path at every size)"""
    from porla_amd import lib
    lib.porla_gpu_set_msm_glv(0 if request.param == "plain" else 1)
    lib.porla_gpu_set_msm_small(0 if request.param == "general" else 1, 0)
    yield request.param
    lib.porla_gpu_set_msm_glv(-1)
    lib.porla_gpu_set_msm_small(1, 0)

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


@pytest.fixture(scope="module")
def mx():
    from porla_amd import multiexp
    return multiexp


@pytest.fixture(scope="module")
def inputs():
    n = 1 << 14
    return common.synth_inputs(n)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 127, 128, 1000, 1408, 3200, 1 << 14])
def test_multi_exp_matches_oracle(mx, inputs, n):
    sc, pt = inputs
    got = mx.bn254_multi_exp(pt[:64 * n], sc[:32 * n], n)
    assert got == common.oracle_msm(sc, pt, n)


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 12, 13, 16, 17, 18, 20])
def test_every_window_width(mx, inputs, c):
    """forces the window width (bucket count, segment length, carries into the top window)"""
    from porla_amd import lib
    sc, pt = inputs
    n = 2000
    lib.porla_gpu_set_msm_window(c)
    try:
        got = mx.msm_host("bn254", sc[:32 * n], pt[:64 * n], n)
    finally:
        lib.porla_gpu_set_msm_window(0)
    assert got == common.oracle_msm(sc, pt, n)


def test_edge_scalars(mx, inputs):
    """zero, one, r-1, r, r+1, 2^256-1 (SetBytes reduction), 2^128: main.go:127"""
    _, pt = inputs
    vals = [0, 1, 2, R - 1, R, R + 1, (1 << 256) - 1, 1 << 128, (1 << 255) + 12345, 5 * R, 5 * R + 7,
            LAMBDA, LAMBDA + 1, LAMBDA - 1, R - LAMBDA, LAMBDA * LAMBDA % R, (R - 1) // 2, (R + 1) // 2, (1 << 126) - 1, 1 << 127,
            147946756881789319000765030803803410728, 9931322734385697763, 3 * LAMBDA % R]
    n = len(vals)
    sc = b"".join(v.to_bytes(32, "big") for v in vals)
    got = mx.bn254_multi_exp(pt[:64 * n], sc, n)
    assert got == common.oracle_msm(sc, pt, n, naive=True)


def test_edge_points(mx, inputs):
    """infinity operands, P and -P pairs (cancel), repeated points in one bucket (P + P)"""
    import bn254_py as o
    sc, pt = inputs
    P0 = pt[:64]
    negP0 = o.neg_point(P0)
    pts = P0 + negP0 + bytes(64) + P0 + P0 + pt[64:128] + bytes(64)
    s7 = (7).to_bytes(32, "big")
    scs = s7 + s7 + sc[:32] + s7 + s7 + sc[32:64] + (0).to_bytes(32, "big")
    n = 7
    got = mx.bn254_multi_exp(pts, scs, n)
    assert got == common.oracle_msm(scs, pts, n, naive=True)
    # everything cancels -> infinity = 64 zero bytes
    got = mx.bn254_multi_exp(P0 + negP0, s7 + s7, 2)
    assert got == bytes(64)


def test_eip196_precompile_vectors_through_the_plugin(mx):
    """external known answers (tests/golden/eip196_kat.json, go-ethereum's alt_bn128 precompile test data) through the
    reference's own symbols: add_point, mult_point and compute_multi_exp"""
    import json
    import os
    kat = json.load(open(os.path.join(common.ROOT, "tests", "golden", "eip196_kat.json")))
    one = (1).to_bytes(32, "big")
    for e in kat["add"]:
        a, b, want = bytes.fromhex(e["a"]), bytes.fromhex(e["b"]), bytes.fromhex(e["sum"])
        assert mx.bn254_add(a, b) == want
        assert mx.bn254_multi_exp(a + b, one + one, 2) == want
    for e in kat["mul"]:
        p, k, want = bytes.fromhex(e["p"]), bytes.fromhex(e["k"]), bytes.fromhex(e["r"])
        assert mx.bn254_mult(p, k) == want
        assert mx.bn254_multi_exp(p, k, 1) == want
        # the same scalar spread over many pairs: k P = sum_i k_i P with sum k_i = k
        parts = [int.from_bytes(k, "big") // 1000] * 999
        parts.append(int.from_bytes(k, "big") - sum(parts))
        sc = b"".join(x.to_bytes(32, "big") for x in parts)
        assert mx.bn254_multi_exp(p * 1000, sc, 1000) == want
    # ... and through the kernels the headline runs (k_bucket_sum30, the reduction tree: more than 32 768 pairs leave the
    # single-launch path) and through the fixed-base commitment kernel, again with geth's expected bytes and no oracle in between
    for e in kat["mul"][:6]:
        p, k, want = bytes.fromhex(e["p"]), bytes.fromhex(e["k"]), bytes.fromhex(e["r"])
        kv = int.from_bytes(k, "big")
        n = 40000
        parts = [kv // n + 3 * i for i in range(n - 1)]
        parts.append(kv - sum(parts))                       # may be negative: taken mod the group order, as fr.SetBytes would see it
        r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
        sc = b"".join((x % r).to_bytes(32, "big") for x in parts)
        assert mx.bn254_multi_exp(p * n, sc, n) == want
        fb = mx.FixedBase("bn254", p, 1)
        try:
            assert fb.commit_host(k, 1, 1) == want
            assert fb.commit_host(sc[:32 * 300], 300, 1)[:64] == mx.bn254_mult(p, sc[:32])
        finally:
            fb.close()


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_reduction_tree_exceptional_operands(mx, inputs, curve):
    """the same point (or a point and its negative) in NEIGHBOURING buckets: the reduction tree then adds equal sums
    (doubling), opposite sums (infinity) and empty nodes at several levels -- the exceptional path of the quad-lane addition"""
    if curve == "bn254":
        import bn254_py as o
        P0 = inputs[1][:64]
        negP0 = o.neg_point(P0)
        oracle = lambda sc, pt, n: common.oracle_msm(sc, pt, n, naive=True)
    else:
        pts = common.secp_bench_points(4)
        P0 = pts[:64]
        pm = 2**256 - 2**32 - 977
        negP0 = P0[:32] + ((pm - int.from_bytes(P0[32:], "big")) % pm).to_bytes(32, "big")
        oracle = lambda sc, pt, n: common.oracle_secp_msm(sc, pt, n)
    be = lambda v: v.to_bytes(32, "big")
    cases = [
        ([1, 2], [P0, P0]),                                 # buckets 0 and 1 hold P: level 0 doubles
        ([1, 2], [P0, negP0]),                              # P and -P: S = infinity, M_0 = -P
        ([1, 2, 3, 4], [P0] * 4),                           # equal sums at levels 0 and 1
        ([1, 2, 3, 4], [P0, negP0, P0, negP0]),
        (list(range(1, 33)), [P0] * 32),                    # every bucket of a 32-bucket block holds P
        (list(range(1, 33)), [P0, negP0] * 16),
        ([5, 5 + (1 << 16), 5 + (1 << 32)], [P0, P0, negP0]),   # the same pattern in three windows
    ]
    for ks, ps in cases:
        sc, pt, n = b"".join(be(k) for k in ks), b"".join(ps), len(ks)
        want = oracle(sc, pt, n)
        assert mx.msm_host(curve, sc, pt, n) == want
        for c in (2, 5, 8):                                  # small windows: more tree levels per bucket index bit
            lib_set(mx, c)
            try:
                assert mx.msm_host(curve, sc, pt, n) == want
            finally:
                lib_set(mx, 0)


def lib_set(mx, c):
    from porla_amd import lib
    lib.porla_gpu_set_msm_window(c)


def test_audit_like_distribution(mx, inputs):
    """abs(int32) coefficients (utils.h:271-275) over 64-way repeated points: the real audit's shape"""
    import random
    sc, pt = inputs
    rnd = random.Random(7)
    n = 3200
    scs = b"".join(mx.bn254_scalar_set_int(rnd.getrandbits(31)) for _ in range(n))
    pts = b"".join(pt[64 * (i % 50):64 * (i % 50) + 64] for i in range(n))
    got = mx.bn254_multi_exp(pts, scs, n)
    assert got == common.oracle_msm(scs, pts, n)


def test_all_same_point_same_scalar(mx, inputs):
    """worst-case skew: every pair lands in the same bucket of every window"""
    sc, pt = inputs
    n = 512
    got = mx.bn254_multi_exp(pt[:64] * n, sc[:32] * n, n)
    assert got == common.oracle_msm(sc[:32] * n, pt[:64] * n, n)


@pytest.mark.parametrize("n,c", [(20000, 0), (20000, 9), (1 << 14, 13), (1 << 14, 16)])
def test_heavy_buckets_are_split_into_work_items(mx, inputs, n, c):
    """skewed inputs: one scalar for all pairs (every window has ONE bucket of n entries), and windows whose top
    digit has 1-2 bits (n = 2^14 -> c = 11: the top window holds all pairs in <= 2 buckets).  The bucket
    accumulation must split such buckets into <= 128-entry work items (k_bucket_sum / k_bucket_combine)."""
    from porla_amd import lib
    sc, pt = inputs
    scs = sc[:32] * n if c in (0, 9) else sc[:32 * n]
    pts = (pt[:64 * 200] * (n // 200 + 1))[:64 * n] if c in (0, 9) else pt[:64 * n]
    lib.porla_gpu_set_msm_window(c)
    try:
        got = mx.msm_host("bn254", scs, pts, n)
    finally:
        lib.porla_gpu_set_msm_window(0)
    assert got == common.oracle_msm(scs, pts, n)


def test_device_pointer_api_and_partials(mx, inputs):
    """porla_bn254_msm_device + range-sharded partial Jacobians folded by porla_bn254_jac_sum"""
    import torch
    sc, pt = inputs
    n = 1 << 14
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    want = common.oracle_msm(sc, pt, n)
    assert mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, stream) == want
    shards = 4
    parts = b""
    for g in range(shards):
        lo, hi = n * g // shards, n * (g + 1) // shards
        parts += mx.msm_device("bn254", d_sc.data_ptr() + 32 * lo, d_pt.data_ptr() + 64 * lo, hi - lo, stream, partial=True)
    assert mx.jac_sum("bn254", parts, shards) == want


def test_kzg_commit_and_open_on_gpu(mx):
    """compute_digest_from_srs / create_proof (main.go:103-116,153-175) vs the Python big-int oracle, and the
    in-reference identity compute_digest(f) == alpha * compute_digest_from_srs(f)"""
    import bn254_py as o
    tau = bytes.fromhex("ffeeddccbbaa99887766554433221100")     # TAU_KEY, config.hpp:39
    alpha = bytes.fromhex("00112233445566778899aabbccddeeff")   # SECRET_KEY, config.hpp:38
    n = 128
    mx.init_key(tau, alpha)
    blob = mx.init_SRS(n)
    assert len(blob) == 32 * n + 132
    data = common.synth_scalars(n, start=1000)
    k = o.KZG(); k.init_key(tau, alpha); k.init_srs(n)
    commit = mx.compute_digest_from_srs(data)
    assert commit == k.compute_digest_from_srs(data)
    assert mx.compute_digest(data) == mx.bn254_mult(commit, alpha.rjust(32, b"\0"))
    c, h, z, y = mx.create_proof(0x1234567890abcdef, data)
    assert (c, h, z, y) == k.create_proof(0x1234567890abcdef, data)
    assert mx.verify_proof(c, h, z, y)
    # server side: SRS from the wire blob
    mx.init_SRS_from_data(n, blob)
    assert mx.compute_digest_from_srs(data) == commit


@pytest.mark.parametrize("dist", ["uniform", "audit"])
def test_full_size_2_20(mx, dist):
    """BASELINE.json config 2: one 2^20-point MSM, bit-exact vs the oracle's multi-threaded bucket MSM"""
    import torch
    n = 1 << 20
    sc, pt = common.cached_inputs(n)
    if dist == "audit":
        import random
        rnd = random.Random(11)
        sc = b"".join((rnd.getrandbits(31)).to_bytes(32, "big") for _ in range(n))
        pt = pt[:64 * (n // 64)] * 64
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    got = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert got == common.oracle_msm(sc, pt, n)


def test_all_ones_limbs_in_montgomery_form(mx):
    """Points whose Montgomery x-coordinate is 0x0ffffffe ffffffff ... ffffffff: in P + P (two equal points in one bucket)
    the doubling squares that coordinate, and every column of the product starts with 0xffffffff * 0xffffffff on top of a
    non-zero incoming accumulator -- the case in which a dropped carry-out of the first v_mad_u64_u32 of a column would
    corrupt the field product (fe_mul_gfx950.inc folds every carry).  The points are on the curve (tools: x = x_mont / R)."""
    pts = [(0x22673664f672e901666cf81e3dcd006df9ca723b2f9061e646da650d1918dddd, 0x249117a44372fa51695e0427975f8c2137758ddab44ad63cb206c1b1b74cefa6),
           (0x2cca58ea976932f57213e21fd33bf2d94a2b2eb52d3f68ab53b289f3e959f687, 0x19d1fe9404fd0b17af06463513e9afaea9dfb1f6ed93892ebc13216977b1ca7d),
           (0x1979bba86b2b7c5bf97116a1c7213b15ada2260c3a521809dfe105078fd8595, 0x1d030e3b5a4b70c13187459682f544b1fc4486820b39f12bcbeb3884c77d476)]
    for x, y in pts:
        P = x.to_bytes(32, "big") + y.to_bytes(32, "big")
        assert common.oracle().oracle_bn254_on_curve(P) == 1
        one = (1).to_bytes(32, "big")
        for k in (2, 3, 7):
            got = mx.bn254_multi_exp(P * k, one * k, k)                 # k equal points, scalar 1: P + P + ...
            assert got == common.oracle_msm(one * k, P * k, k, naive=True)
        sc = (0x1234567).to_bytes(32, "big") * 2
        assert mx.bn254_multi_exp(P * 2, sc, 2) == common.oracle_msm(sc, P * 2, 2, naive=True)
        assert mx.bn254_mult(P, (2).to_bytes(32, "big")) == common.oracle_msm(one * 2, P * 2, 2, naive=True)


def test_scale_2_22_pairs_and_range_partition_invariance(mx):
    """4 Mi pairs (a quarter of BASELINE.json config 3's per-GPU share): random 256-bit scalars over 2^14 distinct points
    repeated 256 times; the result equals the oracle, and folding 4 range-sharded partial sums (the multi-GPU decomposition,
    porla_amd/sharded.py) gives the same point."""
    import torch
    n, distinct = 1 << 22, 1 << 14
    pt = common.synth_points(distinct) * (n // distinct)
    g = torch.Generator(device="cuda").manual_seed(2022)
    d_sc = torch.randint(0, 256, (32 * n,), dtype=torch.uint8, device="cuda", generator=g)
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    got = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, stream)
    parts = b""
    for s in range(4):
        lo = s * (n // 4)
        parts += mx.msm_device("bn254", d_sc.data_ptr() + 32 * lo, d_pt.data_ptr() + 64 * lo, n // 4, stream, partial=True)
    assert mx.jac_sum("bn254", parts, 4) == got
    assert got == common.oracle_msm(bytes(d_sc.cpu().numpy()), pt, n)


def test_two_phase_api_keeps_independent_msms_in_flight(mx, inputs):
    """porla_bn254_msm_device_begin/_end: three MSMs in flight on three streams, retired in order; a slot cannot be begun
    twice; results (affine and Jacobian-partial forms) equal the blocking call"""
    import torch
    from porla_amd import lib
    sc, pt = inputs
    n = 1 << 14
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    streams = [torch.cuda.Stream() for _ in range(3)]
    sizes = [n, 3000, 129]
    want = [common.oracle_msm(sc, pt, m) for m in sizes]
    torch.cuda.synchronize()
    for rep in range(3):
        for k, m in enumerate(sizes):
            mx.msm_begin(1 + k, d_sc.data_ptr(), d_pt.data_ptr(), m, streams[k].cuda_stream)
        assert lib.porla_bn254_msm_device_begin(1, ctypes.c_void_p(d_sc.data_ptr()), ctypes.c_void_p(d_pt.data_ptr()), 10, None) != 0
        for k in range(3):
            if rep == 1:
                part = mx.msm_end(1 + k, partial=True)
                assert mx.jac_sum("bn254", part, 1) == want[k]
            else:
                assert mx.msm_end(1 + k) == want[k]
    assert lib.porla_bn254_msm_device_begin(7, None, None, 0, None) != 0        # slot out of range
    # an end without its begin is an error (it would also be what a thread on ANOTHER device gets for this slot) ...
    out = ctypes.create_string_buffer(64)
    assert lib.porla_bn254_msm_device_end(2, out, 0) == -4
    # ... while the legitimate empty MSM gives the empty sum
    mx.msm_begin(2, d_sc.data_ptr(), d_pt.data_ptr(), 0, streams[0].cuda_stream)
    assert mx.msm_end(2) == bytes(64)


def test_release_workspaces_and_reuse(mx, inputs):
    """porla_gpu_release_msm_workspaces frees the scratch; the next MSM reallocates it and gives the same result; refused while
    a two-phase MSM is pending"""
    import torch
    from porla_amd import lib
    sc, pt = inputs
    n = 3000
    a = mx.msm_host("bn254", sc[:32 * n], pt[:64 * n], n)
    assert lib.porla_gpu_release_msm_workspaces() == 0
    assert mx.msm_host("bn254", sc[:32 * n], pt[:64 * n], n) == a
    d_sc = torch.frombuffer(bytearray(sc[:32 * n]), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt[:64 * n]), dtype=torch.uint8).cuda()
    mx.msm_begin(1, d_sc.data_ptr(), d_pt.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert lib.porla_gpu_release_msm_workspaces() != 0
    assert mx.msm_end(1) == a
    assert lib.porla_gpu_release_msm_workspaces() == 0
