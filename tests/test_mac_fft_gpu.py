"""Parity of the HIP MAC-side encode ("FFT in the exponent", porla_amd/csrc/mac_fft.hip.h) against the oracle, through the
C ABI (GPU box only).  Reference: the MAC halves of Server::CRebuild_Cached, porla/Server/Server.hpp:1523-1536 (init
scaling), :1590-1609 and :1658-1676 (butterflies); client twin porla/Client/Client.hpp:1040-1453.  Bit-exact 64-byte MACs."""
import ctypes
import json
import os

import pytest

from tests import common

pytestmark = pytest.mark.gpu


def oracle_mac(macs, n, curve, part, ws):
    out = ctypes.create_string_buffer(64 * n)
    common.oracle().oracle_icc_mac_crebuild(macs, ctypes.c_size_t(n), 0 if curve == "bn254" else 1, part, ctypes.c_uint64(ws),
                                            out, common.ncpu())
    return out.raw


def macs_for(curve, n):
    if curve == "bn254":
        return common.synth_points(n, start=5000)
    sc = common.secp_bench_scalars(n, start=300)
    out = ctypes.create_string_buffer(64 * n)
    common.oracle().oracle_secp256k1_mul_g_batch(sc, ctypes.c_size_t(n), out, common.ncpu())
    return out.raw


def test_committed_golden_vectors(form):
    from porla_amd import icc
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mac_golden.json")))
    for c in gold["cases"]:
        raw = bytes.fromhex("".join(c["macs"]))
        assert icc.mac_crebuild_host(raw, c["n"], c["curve"], c["write_step"], 0).hex() == "".join(c["X"])
        assert icc.mac_crebuild_host(raw, c["n"], c["curve"], c["write_step"], 1).hex() == "".join(c["Y"])


@pytest.fixture(params=["matrix", "ladder"])
def form(request):
    """both evaluation forms of the network (mac_fft.hip): N commitments against the per-call base / stage-by-stage ladders"""
    from porla_amd import lib
    lib.porla_icc_mac_set_matrix_max(2048 if request.param == "matrix" else 0)
    yield request.param
    lib.porla_icc_mac_set_matrix_max(0)              # the default: no matrix form


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n,ws", [(2, 0), (8, 3), (64, 0), (256, 77)])
def test_matches_oracle(curve, n, ws, form):
    from porla_amd import icc
    macs = macs_for(curve, n)
    for part in (0, 1):
        assert icc.mac_crebuild_host(macs, n, curve, ws, part) == oracle_mac(macs, n, curve, part, ws)


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n", [1024, 4096])
def test_degenerate_inputs_through_every_ladder_kernel(curve, n):
    """round 5's ladders keep the accumulator in registers and leave the register form only for equal-x operands (P + P, P - P) and
    infinity: inputs made of ONE point repeated (stage 1: every um + tm is a doubling, every um - tm infinity; later stages meet
    2^k P against 2^k P and infinity against infinity), of a point alternating with its negative, and of infinity alternating with a
    point -- through the stage-1 kernel, the wave-uniform quad ladder, the eight-lane kernel (per-butterfly scalars) and, with
    PORLA_MAC_QUAD_MAX=0 (tests/test_env_switches_gpu.py), the one-lane kernels"""
    from porla_amd import icc, lib
    lib.porla_icc_mac_set_matrix_max(0)
    try:
        base = macs_for(curve, 2)
        p, q = base[:64], base[64:128]
        neg_p = icc.mac_crebuild_host(bytes(64) + p, 2, curve, 0, 0)[64:128]        # stage 1 of a 2-row network: O - P
        for name, macs in (("one point", p * n), ("P, -P", (p + neg_p) * (n // 2)), ("O, P", (bytes(64) + p) * (n // 2)),
                           ("P, P, Q, Q", (p + p + q + q) * (n // 4))):
            for part in (0, 1):
                assert icc.mac_crebuild_host(macs, n, curve, 5, part) == oracle_mac(macs, n, curve, part, 5), (name, part)
    finally:
        lib.porla_icc_mac_set_matrix_max(0)              # the default: no matrix form


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_infinity_and_repeated_macs(curve, form):
    """fresh levels hold infinity MACs (bn254_set_infinity, Server.hpp:1533-1534); equal MACs exercise P + P and P - P"""
    from porla_amd import icc
    n = 16
    macs = bytearray(macs_for(curve, n))
    macs[64 * 3:64 * 4] = bytes(64)
    macs[64 * 8:64 * 9] = macs[0:64]         # stage 4 pairs rows 0 and 8 with twiddle 1: um + tm = 2P, um - tm = O
    macs[64 * 9:64 * 10] = bytes(64)
    for part in (0, 1):
        got = icc.mac_crebuild_host(bytes(macs), n, curve, 6, part)
        assert got == oracle_mac(bytes(macs), n, curve, part, 6)
    assert icc.mac_crebuild_host(bytes(64 * n), n, curve, 0, 0) == bytes(64 * n)


def test_matches_data_side_network_in_the_exponent():
    """the property the protocol relies on: MAC_i = s_i * G  =>  encoded MAC_k = (data-side encode of s)_k mod q * G,
    checked at N = 1024 through the engine's own data-side encode (porla_icc_encode) and mult_point"""
    import hashlib
    from porla_amd import icc, multiexp as mx
    n = 1024
    s = [int.from_bytes(hashlib.sha256(b"mac-exp" + i.to_bytes(4, "little")).digest(), "big") >> 3 for i in range(n)]
    rows = b"".join(v.to_bytes(32, "little") for v in s)
    x_rows = icc.crebuild_host(rows, n, 1, "bn254", 0, 0, want_aligned=False, want_scalars=False)[0]   # values mod LCM
    sc = b"".join(v.to_bytes(32, "big") for v in s)
    macs = ctypes.create_string_buffer(64 * n)
    common.oracle().oracle_bn254_fixed_base(sc, ctypes.c_size_t(n), macs, common.ncpu())
    got = icc.mac_crebuild_host(macs.raw, n, "bn254", 0, 0)
    R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    G = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    for k in (0, 1, 511, 512, 1023):
        e = int.from_bytes(x_rows[64 * k:64 * k + 64], "little") % R
        assert got[64 * k:64 * k + 64] == mx.bn254_mult(G, e.to_bytes(32, "big"))


def test_device_pointer_api_4096():
    import torch
    from porla_amd import icc
    n = 4096
    macs = macs_for("bn254", n)
    d_in = torch.frombuffer(bytearray(macs), dtype=torch.uint8).cuda()
    d_out = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    icc.mac_crebuild_device(d_in.data_ptr(), n, "bn254", 0, 0, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bytes(d_out.cpu().numpy()) == oracle_mac(macs, n, "bn254", 0, 0)


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n,ws", [(2, 1), (16, 6), (256, 77), (1024, 5), (4096, 123457)])
def test_both_parts_from_one_network_match_the_two_single_part_encodes(curve, n, ws, form):
    """porla_icc_mac_encode_xy_*: the network is linear over Z_q and the Y part is the X part's network on inputs scaled by wt
    (Server.hpp:1494-1536, :1691-1830), so Y_k = wt * X_k -- one scalar multiplication per row instead of a second run of the
    stages.  Both outputs against the oracle's two separate encodes; infinity MACs and repeated MACs among the inputs."""
    from porla_amd import icc
    macs = bytearray(macs_for(curve, n))
    if n >= 16:
        macs[64 * 3:64 * 4] = bytes(64)
        macs[64 * 8:64 * 9] = macs[0:64]
        macs[64 * 9:64 * 10] = bytes(64)
    macs = bytes(macs)
    x, y = icc.mac_crebuild_xy_host(macs, n, curve, ws)
    assert x == oracle_mac(macs, n, curve, 0, ws)
    assert y == oracle_mac(macs, n, curve, 1, ws)
    assert icc.mac_crebuild_xy_host(bytes(64 * n), n, curve, ws) == (bytes(64 * n), bytes(64 * n))


def test_both_parts_device_form_at_2_15_rows():
    """the size of a large CRebuild (one lane per row for the scaling above 2^14 rows), device pointers, against the oracle"""
    import torch
    from porla_amd import icc
    n = 1 << 15
    macs = macs_for("bn254", n)
    d_in = torch.frombuffer(bytearray(macs), dtype=torch.uint8).cuda()
    d_x = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    d_y = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    icc.mac_crebuild_xy_device(d_in.data_ptr(), n, "bn254", 1000003, d_x.data_ptr(), d_y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bytes(d_x.cpu().numpy()) == oracle_mac(macs, n, "bn254", 0, 1000003)
    assert bytes(d_y.cpu().numpy()) == oracle_mac(macs, n, "bn254", 1, 1000003)
