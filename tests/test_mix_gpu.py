"""Parity of the HIP Server::mix kernels (data part porla_icc_mix_*, MAC part porla_icc_mac_mix_*) against the oracle
(oracle/icc_ref.c:oracle_icc_mix, oracle/mac_ref.c:oracle_icc_mac_mix, both pinned to the Python restatements of
porla/Server/Server.hpp:1209-1328).  Bit-exact."""
import ctypes
import random

import pytest

from tests import common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("length,ncols,n_total", [(1, 128, 2), (4, 5, 64), (256, 128, 1024), (1024, 16, 1024)])
def test_data_mix(curve, length, ncols, n_total):
    import icc_py
    from porla_amd import icc
    rnd = random.Random(length * 131 + ncols)
    lcm = icc_py.LCM[curve]
    vals0 = [rnd.randrange(lcm) for _ in range(length * ncols)]
    vals1 = [rnd.randrange(lcm) for _ in range(length * ncols)]
    vals0[0], vals1[0] = lcm - 1, lcm - 1
    vals0[-1], vals1[-1] = 0, 0
    a0 = b"".join(v.to_bytes(64, "little") for v in vals0)
    a1 = b"".join(v.to_bytes(64, "little") for v in vals1)
    got = icc.mix_host(a0, a1, length, ncols, n_total, curve)
    want = ctypes.create_string_buffer(2 * length * ncols * 64)
    common.oracle().oracle_icc_mix(a0, a1, ctypes.c_size_t(length), ctypes.c_size_t(ncols), ctypes.c_size_t(n_total),
                                   icc.CURVE[curve], want)
    assert got == want.raw
    if length <= 4:     # and directly against the Python restatement
        rows0 = [vals0[i * ncols:(i + 1) * ncols] for i in range(length)]
        rows1 = [vals1[i * ncols:(i + 1) * ncols] for i in range(length)]
        py = icc_py.mix(rows0, rows1, n_total, curve)
        assert got == b"".join(v.to_bytes(64, "little") for r in py for v in r)


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("length,n_total", [(1, 2), (8, 64), (512, 1024)])
def test_mac_mix(curve, length, n_total):
    from porla_amd import icc
    from tests.test_mac_fft_gpu import macs_for
    macs = bytearray(macs_for(curve, 2 * length))
    if length >= 8:
        macs[64 * 2:64 * 3] = bytes(64)                                    # infinity in A0
        macs[64 * (length + 3):64 * (length + 4)] = bytes(64)              # infinity in A1
    a0, a1 = bytes(macs[:64 * length]), bytes(macs[64 * length:])
    got = icc.mac_mix_host(a0, a1, length, n_total, curve)
    want = ctypes.create_string_buffer(2 * length * 64)
    common.oracle().oracle_icc_mac_mix(a0, a1, ctypes.c_size_t(length), ctypes.c_size_t(n_total), icc.CURVE[curve], want,
                                       common.ncpu())
    assert got == want.raw


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("length,n_total", [(64, 64), (2048, 1 << 15), (16384, 1 << 15)])
def test_mac_mix_of_a_block_with_itself_and_with_infinity(curve, length, n_total):
    """the equal-x exits of the register-form additions: A1 = A0 (row 0 multiplies by v^0 = 1: A0 + A0 is a doubling, A0 - A0 infinity),
    A0 = infinity throughout, and both blocks one repeated point -- at lengths that take the eight-lane kernel (<= 2^13 elements), the
    four-lane one (2^14) and, with n_total = length, a twiddle step of one"""
    from porla_amd import icc
    from tests.test_mac_fft_gpu import macs_for
    a0 = macs_for(curve, min(length, 256)) * (length // min(length, 256))
    one = a0[:64] * length
    for name, x0, x1 in (("A1 = A0", a0, a0), ("A0 = O", bytes(64 * length), a0), ("one point", one, one)):
        got = icc.mac_mix_host(x0, x1, length, n_total, curve)
        want = ctypes.create_string_buffer(2 * length * 64)
        common.oracle().oracle_icc_mac_mix(x0, x1, ctypes.c_size_t(length), ctypes.c_size_t(n_total), icc.CURVE[curve], want,
                                           common.ncpu())
        assert got == want.raw, name


def test_mac_mix_with_an_empty_upper_block():
    """algebraic identity: mixing A0 with a block of infinity MACs (a fresh level, Server.hpp:1533-1534) returns A0 twice"""
    from porla_amd import icc
    from tests.test_mac_fft_gpu import macs_for
    length = 16
    a0 = macs_for("bn254", length)
    got = icc.mac_mix_host(a0, bytes(64 * length), length, 64, "bn254")
    assert got == a0 + a0


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("length,n_total", [(1, 2), (64, 1024), (8192, 16384), (16384, 1 << 17)])
def test_mac_mix_pair_in_one_launch(curve, length, n_total):
    """porla_icc_mac_mix_pair_device: Server::mix's butterfly on the MAC commitments and on the MAC alignments (same v^i,
    Server.hpp:1281-1318) in one launch -- each output against the oracle (up to 2^13 butterflies in total the four-lane kernel,
    above it one lane per butterfly)"""
    import torch
    from porla_amd import icc, lib
    from tests.test_mac_fft_gpu import macs_for
    pool = macs_for(curve, min(4 * length, 2048))
    arr = bytearray((pool * (4 * length * 64 // len(pool) + 1))[:64 * 4 * length])
    if length >= 64:
        arr[64 * 5:64 * 6] = bytes(64)                                     # infinity in the first array's A0
        arr[64 * (3 * length + 7):64 * (3 * length + 8)] = bytes(64)       # ... and in the second array's A1
    parts = [bytes(arr[64 * length * k:64 * length * (k + 1)]) for k in range(4)]
    d = [torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda() for p in parts]
    oa, ob = (torch.empty(128 * length, dtype=torch.uint8, device="cuda") for _ in range(2))
    vp = ctypes.c_void_p
    rc = lib.porla_icc_mac_mix_pair_device(vp(d[0].data_ptr()), vp(d[1].data_ptr()), vp(d[2].data_ptr()), vp(d[3].data_ptr()), length, n_total,
                                           icc.CURVE[curve], vp(oa.data_ptr()), vp(ob.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    for out, (a0, a1) in ((oa, parts[0:2]), (ob, parts[2:4])):
        want = ctypes.create_string_buffer(2 * length * 64)
        common.oracle().oracle_icc_mac_mix(a0, a1, ctypes.c_size_t(length), ctypes.c_size_t(n_total), icc.CURVE[curve], want, common.ncpu())
        assert bytes(out.cpu().numpy()) == want.raw


@pytest.mark.parametrize("curve,length,n_total", [("bn254", 32, 256), ("secp256k1", 8, 64), ("bn254", 4096, 16384)])
def test_server_mix_in_one_call(curve, length, n_total):
    """porla_server_mix_device = Server::mix(is_x, level) (Server.hpp:1209-1328): data rows, MAC commitments and MAC alignments of
    one mix from one call (two streams inside) -- each output against its oracle"""
    import torch
    import icc_py
    from porla_amd import icc, lib
    from tests.test_mac_fft_gpu import macs_for
    ncols = 128 if length <= 64 else 8
    rnd = random.Random(length + n_total)
    lcm = icc_py.LCM[curve]
    a0 = b"".join(rnd.randrange(lcm).to_bytes(64, "little") for _ in range(length * ncols))
    a1 = b"".join(rnd.randrange(lcm).to_bytes(64, "little") for _ in range(length * ncols))
    pool = macs_for(curve, min(4 * length, 2048))
    arr = (pool * (4 * length * 64 // len(pool) + 1))[:64 * 4 * length]
    parts = [arr[64 * length * k:64 * length * (k + 1)] for k in range(4)]
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    d = [dev(a0), dev(a1)] + [dev(p) for p in parts]
    o_data = torch.empty(2 * length * ncols * 64, dtype=torch.uint8, device="cuda")
    o_mac, o_al = (torch.empty(128 * length, dtype=torch.uint8, device="cuda") for _ in range(2))
    vp = ctypes.c_void_p
    rc = lib.porla_server_mix_device(*[vp(t.data_ptr()) for t in d], length, ncols, n_total, icc.CURVE[curve], vp(o_data.data_ptr()),
                                     vp(o_mac.data_ptr()), vp(o_al.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    want = ctypes.create_string_buffer(2 * length * ncols * 64)
    common.oracle().oracle_icc_mix(a0, a1, ctypes.c_size_t(length), ctypes.c_size_t(ncols), ctypes.c_size_t(n_total), icc.CURVE[curve], want)
    assert bytes(o_data.cpu().numpy()) == want.raw
    for out, (p0, p1) in ((o_mac, parts[0:2]), (o_al, parts[2:4])):
        w = ctypes.create_string_buffer(2 * length * 64)
        common.oracle().oracle_icc_mac_mix(p0, p1, ctypes.c_size_t(length), ctypes.c_size_t(n_total), icc.CURVE[curve], w, common.ncpu())
        assert bytes(out.cpu().numpy()) == w.raw
