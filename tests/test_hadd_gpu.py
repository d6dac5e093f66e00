"""GPU box: Server::HAdd / Client::HAdd and the HRebuild chains (porla/Server/Server.hpp:1388-1477, 1329-1386;
porla/Client/Client.hpp:978-1038) through their named C-ABI wrappers, against the Python restatement oracle/icc_py.py:hadd /
hrebuild (loop for loop) -- bit-exact, both curves."""
import ctypes
import random

import pytest

from tests import common

pytestmark = pytest.mark.gpu
TAU = bytes.fromhex("ffeeddccbbaa99887766554433221100")
ALPHA = bytes.fromhex("00112233445566778899aabbccddeeff")


def pt_bytes(p):
    return bytes(64) if p is None else p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")


def pt_tuple(b):
    return None if b == bytes(64) else (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))


def some_points(curve, n):
    if curve == "bn254":
        raw = common.synth_points(n)
    else:
        raw = common.secp_bench_points(n)
    return [raw[64 * i:64 * i + 64] for i in range(n)]


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n_total,write_step", [(2, 0), (16, 5), (1024, 777), (1 << 15, 123456789)])
def test_hadd_data_and_mac_parts(curve, n_total, write_step):
    import icc_py
    from porla_amd import icc
    rnd = random.Random(n_total + write_step)
    ncols = 128
    vals = [rnd.getrandbits(256) for _ in range(ncols)]
    vals[0], vals[1], vals[2] = 0, (1 << 256) - 1, icc_py.P_ICC      # chunk values are any 256-bit integers
    data = b"".join(v.to_bytes(32, "little") for v in vals)
    mac = some_points(curve, 3)[2]
    mods, cs, mac_b2, wt = icc_py.hadd(vals, pt_tuple(mac), n_total, write_step, curve)
    b2, sc, wts = icc.hadd_host(data, n_total, write_step, curve)
    assert b2 == b"".join(m.to_bytes(32, "little") for m in mods)
    assert sc == b"".join(c.to_bytes(32, "big") for c in cs)
    assert wts == wt.to_bytes(32, "big")
    _, sc_le, _ = icc.hadd_host(data, n_total, write_step, curve, scalar_le=True)
    assert sc_le == b"".join(c.to_bytes(32, "little") for c in cs)
    assert icc.mac_scale_host(mac, n_total, write_step, curve) == pt_bytes(mac_b2)
    assert icc.mac_scale_host(bytes(64), n_total, write_step, curve) == bytes(64)


def test_kzg_hadd_all_three_outputs():
    """the KZG build's HAdd: MAC_align_B2 is the commitment of the alignment scalars to the SRS (align_MAC -> compute_digest_from_srs)"""
    import icc_py
    from porla_amd import icc, multiexp as mx
    mx.init_key(TAU, ALPHA)
    mx.init_SRS_from_data(128, mx.init_SRS(128))
    rnd = random.Random(42)
    vals = [rnd.getrandbits(256) for _ in range(128)]
    data = b"".join(v.to_bytes(32, "little") for v in vals)
    mac = mx.compute_digest(b"".join(v.to_bytes(32, "big") for v in vals))           # the block's MAC as the client makes it
    n_total, ws = 1024, 313
    mods, cs, mac_b2, _ = icc_py.hadd(vals, pt_tuple(mac), n_total, ws, "bn254")
    b2, m2, ma = icc.kzg_hadd_host(data, mac, n_total, ws)
    assert b2 == b"".join(m.to_bytes(32, "little") for m in mods)
    assert m2 == pt_bytes(mac_b2)
    assert ma == mx.compute_digest_from_srs(b"".join(c.to_bytes(32, "big") for c in cs))
    # what align_MAC is for: MAC_B2 + MAC_align_B2 is the MAC of the ALIGNED block (commitments are linear in the chunks mod r)
    aligned_be = b"".join(m.to_bytes(32, "big") for m in mods)
    a = ctypes.create_string_buffer(64)
    want = mx.bn254_mult(mx.compute_digest_from_srs(aligned_be), ALPHA.rjust(32, b"\0"))
    srs_mac = mx.bn254_mult(mx.compute_digest_from_srs(b"".join(v.to_bytes(32, "big") for v in vals)), ALPHA.rjust(32, b"\0"))
    assert srs_mac == mac                                                           # compute_digest == alpha * compute_digest_from_srs
    del a, want


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("level,n_total,ncols", [(0, 16, 8), (1, 16, 8), (3, 16, 128), (6, 64, 16), (5, 1024, 128)])
def test_hrebuild_chains(curve, level, n_total, ncols):
    import icc_py
    from porla_amd import icc
    rnd = random.Random(level * 1000 + ncols)
    lcm = icc_py.LCM[curve]
    pts = some_points(curve, 8)
    # levels[i]: 2 * 2^i rows; resident halves + level 0's incoming row are meaningful, the rest is overwritten
    rows = [[[rnd.randrange(lcm) for _ in range(ncols)] for _ in range(2 << i)] for i in range(level + 1)]
    macs = [[pt_tuple(pts[rnd.randrange(8)]) if rnd.random() < 0.9 else None for _ in range(2 << i)] for i in range(level + 1)]
    data_bufs = [ctypes.create_string_buffer(b"".join(v.to_bytes(64, "little") for r in lv for v in r), (2 << i) * ncols * 64)
                 for i, lv in enumerate(rows)]
    mac_bufs = [ctypes.create_string_buffer(b"".join(pt_bytes(p) for p in lv), (2 << i) * 64) for i, lv in enumerate(macs)]
    icc_py.hrebuild(rows, level, n_total, curve)
    icc_py.hrebuild(macs, level, n_total, curve, mac=True)
    icc.hrebuild_host(data_bufs, level, n_total, curve, n_cols=ncols)
    icc.mac_hrebuild_host(mac_bufs, level, n_total, curve)
    for i in range(level + 1):
        assert data_bufs[i].raw == b"".join(v.to_bytes(64, "little") for r in rows[i] for v in r), i
        assert mac_bufs[i].raw == b"".join(pt_bytes(p) for p in macs[i]), i


def test_hrebuild_equals_the_sequence_of_mix_calls():
    """the one-call chain is what `level` separate mix calls (the entry points of tests/test_mix_gpu.py) give"""
    import icc_py
    from porla_amd import icc
    rnd = random.Random(9)
    level, n_total, ncols = 4, 256, 128
    lcm = icc_py.LCM["bn254"]
    raw = [bytes().join(rnd.randrange(lcm).to_bytes(64, "little") for _ in range((2 << i) * ncols)) for i in range(level + 1)]
    bufs = [ctypes.create_string_buffer(r, len(r)) for r in raw]
    icc.hrebuild_host(bufs, level, n_total, "bn254")
    cur = raw[0][ncols * 64:2 * ncols * 64]
    for i in range(level):
        ln = 1 << i
        cur = icc.mix_host(raw[i][:ln * ncols * 64], cur, ln, ncols, n_total, "bn254")
    top = 1 << level
    assert bufs[level].raw[:top * ncols * 64] == cur and bufs[level].raw[top * ncols * 64:] == cur
