"""GPU box: the single-launch MSM of msm_small.hip.h (n <= 4096 pairs -- every size the reference issues: the audit's n_points
<= 3200 with abs(int32) coefficients, Server.hpp:585-587, 617-621, 838-848, 900-901; the client's 16 / 176 / 1408-point calls,
Client.hpp:374-406, 756-787) against the oracle, through compute_multi_exp and the device-pointer entries, on both curves:
every window width, every scalar length class (the kernel derives its shape from the scalars' bit length), edge operands."""
import ctypes
import random

import pytest

from tests import common

pytestmark = pytest.mark.gpu
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


@pytest.fixture(scope="module")
def mx():
    from porla_amd import multiexp
    return multiexp


@pytest.fixture(scope="module")
def inputs():
    return common.synth_inputs(4096)


@pytest.fixture(autouse=True)
def small_on():
    from porla_amd import lib
    lib.porla_gpu_set_msm_small(1, 0)
    lib.porla_gpu_set_msm_glv(-1)
    yield
    lib.porla_gpu_set_msm_small(1, 0)


def scalars_of_bits(n, bits, seed):
    rnd = random.Random(seed)
    return b"".join(rnd.getrandbits(bits).to_bytes(32, "big") for _ in range(n))


@pytest.mark.parametrize("c", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("n", [1, 7, 128, 1408, 3200, 4096])
def test_every_window_width_full_scalars(mx, inputs, c, n):
    from porla_amd import lib
    sc, pt = inputs
    lib.porla_gpu_set_msm_small(1, c)
    got = mx.bn254_multi_exp(pt[:64 * n], sc[:32 * n], n)
    assert got == common.oracle_msm(sc, pt, n)
    cc, W, glv = mx.last_msm_shape()
    assert glv and (c == 0 or cc == c) and W == (126 + 1 + cc - 1) // cc     # 256-bit SHA scalars: split into 126-bit halves


@pytest.mark.parametrize("bits", [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 249, 250, 253, 254, 255, 256])
def test_every_scalar_length_class(mx, inputs, bits):
    """the kernel ORs the scalars and sizes its windows from the result: short scalars -> few windows, no split;
    > 128 bits -> endomorphism split; >= 250 bits -> reduction mod r first (fr.SetBytes, main.go:127)"""
    _, pt = inputs
    for n in (3, 500, 3200):
        sc = scalars_of_bits(n, bits, bits * 1000 + n)
        # make sure the longest scalar really has `bits` bits
        sc = ((1 << (bits - 1)) | 1).to_bytes(32, "big") + sc[32:]
        assert mx.bn254_multi_exp(pt[:64 * n], sc, n) == common.oracle_msm(sc, pt, n)
        cc, W, glv = mx.last_msm_shape()
        assert glv == (bits > 128)
        if bits <= 128:
            assert W == (bits + 1 + cc - 1) // cc


def test_audit_shape_abs_int32(mx, inputs):
    """abs(int32) coefficients (bn254_scalar_set_int, utils.h:271-275) over repeated points: 31 bits of windows, whatever their width"""
    _, pt = inputs
    rnd = random.Random(11)
    for n in (128, 1408, 3200):
        sc = b"".join(mx.bn254_scalar_set_int(rnd.getrandbits(31)) for _ in range(n))
        pts = b"".join(pt[64 * (i % 50):64 * (i % 50) + 64] for i in range(n))
        assert mx.bn254_multi_exp(pts, sc, n) == common.oracle_msm(sc, pts, n)
        cc, W, glv = mx.last_msm_shape()
        assert not glv and W * cc <= 32 + cc          # windows cover the 31 bits (+ carry), not 254


def test_edge_operands(mx, inputs):
    import bn254_py as o
    sc, pt = inputs
    P0 = pt[:64]
    neg = o.neg_point(P0)
    be = lambda v: v.to_bytes(32, "big")
    cases = [
        ([0], [P0]), ([0, 0, 0], [P0, pt[64:128], neg]),                      # all-zero scalars -> infinity
        ([5, 5], [P0, neg]),                                                  # cancels inside one bucket
        ([5, 5], [P0, P0]),                                                   # the bucket doubles
        ([7, 7, 7, 7, 7], [P0] * 5),
        ([1, 2, 3, 4, 5, 6, 7, 8], [P0] * 8), ([1, 2, 3, 4, 5, 6, 7, 8], [P0, neg] * 4),   # equal / opposite sums in neighbouring buckets
        ([R - 1, R, R + 1, (1 << 256) - 1, 5 * R + 3], [P0, pt[64:128], pt[128:192], pt[192:256], pt[256:320]]),
        ([3, 9, 27], [bytes(64), P0, bytes(64)]),                             # infinity points
        ([1 << 128, (1 << 128) - 1, 1 << 127], [P0, P0, neg]),
    ]
    for ks, ps in cases:
        s, p, n = b"".join(be(k) for k in ks), b"".join(ps), len(ks)
        want = common.oracle_msm(s, p, n, naive=True)
        for c in (0, 1, 2, 8):
            from porla_amd import lib
            lib.porla_gpu_set_msm_small(1, c)
            assert mx.bn254_multi_exp(p, s, n) == want, (ks, c)
    n = 512                                                                   # one bucket takes everything
    assert mx.bn254_multi_exp(pt[:64] * n, sc[:32] * n, n) == common.oracle_msm(sc[:32] * n, pt[:64] * n, n)


def test_general_and_single_launch_paths_agree(mx, inputs):
    from porla_amd import lib
    sc, pt = inputs
    for n in (2, 999, 4096):
        lib.porla_gpu_set_msm_small(1, 0)
        a = mx.msm_host("bn254", sc[:32 * n], pt[:64 * n], n)
        lib.porla_gpu_set_msm_small(0, 0)
        b = mx.msm_host("bn254", sc[:32 * n], pt[:64 * n], n)
        assert a == b == common.oracle_msm(sc, pt, n)


@pytest.mark.parametrize("c", [0, 1, 2, 5, 8])
def test_secp256k1(mx, c):
    from porla_amd import lib
    lib.porla_gpu_set_msm_small(1, c)
    for n in (1, 16, 176, 1408, 4096):
        sc, pt = common.secp_bench_scalars(n), common.secp_bench_points(n)
        want = common.oracle_secp_msm(sc, pt, n)
        assert mx.msm_host("secp256k1", sc, pt, n) == want
        assert want == common.secp_bench_expected(sc, n)
    n = 1408                                                                  # the IPA audit: abs(int32) coefficients
    rnd = random.Random(3)
    sc = b"".join(rnd.getrandbits(31).to_bytes(32, "big") for _ in range(n))
    pt = common.secp_bench_points(n)
    assert mx.msm_host("secp256k1", sc, pt, n) == common.oracle_secp_msm(sc, pt, n)
    # scalars around the group order and 2^256 - 1 (no reduction happens on this curve below n; above it one subtraction)
    N = common.SECP_N
    vals = [N - 1, N, N + 1, (1 << 256) - 1, 1 << 255, (1 << 128) + 1]
    sc = b"".join(v.to_bytes(32, "big") for v in vals)
    pt = common.secp_bench_points(len(vals))
    assert mx.msm_host("secp256k1", sc, pt, len(vals)) == common.oracle_secp_msm(sc, pt, len(vals), naive=True)


def test_two_msms_in_flight_do_not_block_each_other(mx, inputs):
    """the audit issues its MSMs in pairs (Server.hpp:900-901): begin() of the single-launch path returns without a host
    round trip (no blocking pre-scan), so both are enqueued before either is waited for"""
    import time
    import torch
    sc, pt = inputs
    n = 3200
    rnd = random.Random(5)
    a_sc = b"".join(mx.bn254_scalar_set_int(rnd.getrandbits(31)) for _ in range(n))
    d_a = torch.frombuffer(bytearray(a_sc), dtype=torch.uint8).cuda()
    d_sc = torch.frombuffer(bytearray(sc[:32 * n]), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt[:64 * n]), dtype=torch.uint8).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    want = [common.oracle_msm(a_sc, pt, n), common.oracle_msm(sc, pt, n)]
    for _ in range(3):
        t0 = time.perf_counter()
        mx.msm_begin(1, d_a.data_ptr(), d_pt.data_ptr(), n, s1.cuda_stream)
        mx.msm_begin(2, d_sc.data_ptr(), d_pt.data_ptr(), n, s2.cuda_stream)
        t_enqueue = time.perf_counter() - t0
        assert mx.msm_end(1) == want[0]
        assert mx.msm_end(2) == want[1]
    assert t_enqueue < 2e-3


@pytest.mark.parametrize("c", [0, 1, 4, 8])
def test_audit_pair_one_launch(mx, inputs, c):
    """Server::audit ends in TWO MSMs over the same coefficients (combined_MAC over the commitments, combined_align over the
    alignment points; Server.hpp:900-901): porla_bn254_msm_pair_* runs both in one launch -- each must equal its own MSM"""
    import torch
    from porla_amd import lib
    sc, pt = inputs
    lib.porla_gpu_set_msm_small(1, c)
    rnd = random.Random(17)
    for n in (1, 2, 128, 1408, 3200, 4096):
        for kind in ("int32", "full"):
            s = b"".join(mx.bn254_scalar_set_int(rnd.getrandbits(31)) for _ in range(n)) if kind == "int32" else sc[:32 * n]
            pa = pt[:64 * n]
            pb = b"".join(pt[64 * ((7 * i + 3) % 4096):64 * ((7 * i + 3) % 4096) + 64] for i in range(n))
            want = (common.oracle_msm(s, pa, n), common.oracle_msm(s, pb, n))
            assert mx.msm_pair_host("bn254", s, pa, pb, n) == want, (n, kind)
            if n in (128, 3200):
                d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in (s, pa, pb)]
                torch.cuda.synchronize()
                for _ in range(3):       # repeated launches on one workspace: the arrival counters of both halves reset themselves
                    assert mx.msm_pair_device("bn254", d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n) == want
                assert mx.msm_device("bn254", d[0].data_ptr(), d[2].data_ptr(), n) == want[1]   # and the lone form still works after a pair
    # one set sums to infinity, the other does not; and n = 0
    import bn254_py as o
    P0 = pt[:64]
    s = (5).to_bytes(32, "big") * 2
    assert mx.msm_pair_host("bn254", s, P0 + o.neg_point(P0), P0 + P0, 2) == (bytes(64), common.oracle_msm(s, P0 + P0, 2, naive=True))
    assert mx.msm_pair_host("bn254", b"", b"", b"", 0) == (bytes(64), bytes(64))


def test_audit_pair_secp256k1_and_large(mx, inputs):
    """the IPA twin (Server.hpp:842-848) and the fall-through above 32 768 pairs (two MSMs, one after the other)"""
    rnd = random.Random(23)
    for n in (16, 1408):
        s = b"".join(rnd.getrandbits(31).to_bytes(32, "big") for _ in range(n))
        pa = common.secp_bench_points(n)
        pb = pa[64:] + pa[:64]
        assert mx.msm_pair_host("secp256k1", s, pa, pb, n) == (common.oracle_secp_msm(s, pa, n), common.oracle_secp_msm(s, pb, n))
    sc, pt = common.synth_inputs(8192)
    for n in (8192, 40000):               # one launch up to 32 768 pairs; above: the general path twice (points repeated to get there)
        s, pa = (sc * 5)[:32 * n], (pt * 5)[:64 * n]
        pb = pa[64 * 100:] + pa[:64 * 100]
        assert mx.msm_pair_host("bn254", s, pa, pb, n) == (common.oracle_msm(s, pa, n), common.oracle_msm(s, pb, n))


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_audit_pair_from_resident_stores(mx, inputs, curve):
    """porla_*_audit_msm_pair_device: the gather Server::audit does on the host (ptc / pta / sc from the challenged indices,
    Server.hpp:838-848, 893-899) on the device, then the pair MSM -- against the oracle on the host-gathered arrays"""
    import numpy as np
    import torch
    store = 512
    if curve == "bn254":
        pts = inputs[1][:64 * store]
        oracle = common.oracle_msm
    else:
        pts = common.secp_bench_points(store)
        oracle = common.oracle_secp_msm
    a = bytearray(pts)
    a[64 * 7:64 * 8] = bytes(64)                                  # an infinity entry in store a
    a = bytes(a)
    b = pts[64 * 100:] + pts[:64 * 100]
    d_a = torch.frombuffer(bytearray(a), dtype=torch.uint8).cuda()
    d_b = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    rnd = random.Random(31)
    for n in (1, 128, 1408, 3200):
        idx = [rnd.randrange(store) for _ in range(n)]
        coef = [rnd.getrandbits(31) for _ in range(n)]
        if n > 2:
            idx[0], idx[1], coef[0], coef[1] = 7, 7, 0x7fffffff, 0x80000000        # abs(INT_MIN) read back as unsigned
        d_idx = torch.tensor(idx, dtype=torch.int64).cuda()
        d_coef = torch.tensor(np.array(coef, dtype=np.uint32).view(np.int32)).cuda()
        torch.cuda.synchronize()
        sc = b"".join(c.to_bytes(32, "big") for c in coef)
        pa = b"".join(a[64 * i:64 * i + 64] for i in idx)
        pb = b"".join(b[64 * i:64 * i + 64] for i in idx)
        got = mx.audit_msm_pair_device(curve, d_a.data_ptr(), d_b.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), n)
        assert got == (oracle(sc, pa, n), oracle(sc, pb, n)), (curve, n)


def test_audit_pair_two_phase_and_commit_to_host(mx, inputs):
    """begin / end of the audit pair with other work in between; a begun pair is not collected by the single-MSM end; and
    porla_kzg_commit_batch_device_to_host == compute_digest_from_srs row by row (1 row: one launch; 100 rows: batch kernels)"""
    import numpy as np
    import torch
    from porla_amd import lib
    store = 256
    pts = inputs[1][:64 * store]
    b = pts[64 * 9:] + pts[:64 * 9]
    d_a = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    d_b = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    rnd = random.Random(41)
    n = 1408
    idx = [rnd.randrange(store) for _ in range(n)]
    coef = [rnd.getrandbits(31) for _ in range(n)]
    d_idx = torch.tensor(idx, dtype=torch.int64).cuda()
    d_coef = torch.tensor(np.array(coef, dtype=np.uint32).view(np.int32)).cuda()
    torch.cuda.synchronize()
    sc = b"".join(c.to_bytes(32, "big") for c in coef)
    want = (common.oracle_msm(sc, b"".join(pts[64 * i:64 * i + 64] for i in idx), n),
            common.oracle_msm(sc, b"".join(b[64 * i:64 * i + 64] for i in idx), n))
    s2 = torch.cuda.Stream()
    for _ in range(2):
        mx.audit_msm_pair_begin(2, "bn254", d_a.data_ptr(), d_b.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), n, s2.cuda_stream)
        assert mx.bn254_multi_exp(pts[:64 * 200], inputs[0][:32 * 200], 200) == common.oracle_msm(inputs[0], pts, 200)   # slot 0 meanwhile
        out = ctypes.create_string_buffer(64)
        assert lib.porla_bn254_msm_device_end(2, out, 0) != 0                  # a pair is pending there, not a single MSM
        assert mx.audit_msm_pair_end(2, "bn254") == want
    with pytest.raises(RuntimeError):
        mx.audit_msm_pair_end(2, "bn254")                                      # nothing begun
    mx.init_key(bytes(range(16)), bytes(range(16, 32)))
    mx.init_SRS(128)
    for rows in (1, 3, 100):
        data = bytes(rnd.getrandbits(8) for _ in range(4096 * rows))
        d_rows = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        got = mx.kzg_commit_batch_device_to_host(d_rows.data_ptr(), rows, torch.cuda.current_stream().cuda_stream)
        for r in (0, rows - 1):
            assert got[64 * r:64 * r + 64] == mx.compute_digest_from_srs(data[4096 * r:4096 * r + 4096])
