"""CPU: pins the MAC-side encode oracle.  The C restatement (oracle/mac_ref.c) of the MAC halves of
Server::CRebuild_Cached (porla/Server/Server.hpp:1523-1536, 1590-1609, 1658-1676) against the loop-for-loop Python
restatement (oracle/icc_py.py:mac_crebuild), plus linearity and the 'exponent' consistency with the data-side network:
encoding the MACs s_i * G must give (network applied to the scalars s_i mod q) * G."""
import ctypes

import pytest

from tests import common

GX = {"bn254": (1, 2),
      "secp256k1": (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
                    0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)}


def pt_bytes(p):
    return bytes(64) if p is None else p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")


def c_mac_crebuild(macs_bytes, n, curve, part, write_step):
    out = ctypes.create_string_buffer(64 * n)
    common.oracle().oracle_icc_mac_crebuild(macs_bytes, ctypes.c_size_t(n), 0 if curve == "bn254" else 1, part,
                                            ctypes.c_uint64(write_step), out, common.ncpu())
    return out.raw


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n,ws", [(2, 0), (4, 1), (8, 5), (16, 0)])
def test_c_vs_python(curve, n, ws):
    import icc_py
    macs = [icc_py.ec_mul(curve, GX[curve], 1000 + 37 * i) for i in range(n)]
    macs[n // 2] = None                     # an infinity MAC (bn254_set_infinity, Server.hpp:1538-1539 style)
    if n >= 4:
        macs[3] = macs[0]                   # repeated point
    X, Y = icc_py.mac_crebuild(macs, curve, ws)
    raw = b"".join(pt_bytes(p) for p in macs)
    assert c_mac_crebuild(raw, n, curve, 0, ws) == b"".join(pt_bytes(p) for p in X)
    assert c_mac_crebuild(raw, n, curve, 1, ws) == b"".join(pt_bytes(p) for p in Y)


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_exponent_consistency_with_data_network(curve):
    """MAC_i = s_i * G  ->  encoded MAC_k = (data network applied to s, reduced mod q)_k * G"""
    import icc_py
    n = 8
    s = [(12345 + 7919 * i) for i in range(n)]
    macs = [icc_py.ec_mul(curve, GX[curve], v) for v in s]
    Xd, Yd = icc_py.crebuild([[v] for v in s], curve, 3)
    Xm, Ym = icc_py.mac_crebuild(macs, curve, 3)
    for k in range(n):
        assert Xm[k] == icc_py.ec_mul(curve, GX[curve], Xd[k][0] % icc_py.Q[curve])
        assert Ym[k] == icc_py.ec_mul(curve, GX[curve], Yd[k][0] % icc_py.Q[curve])


def test_c_oracle_matches_committed_golden():
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mac_golden.json")))
    for c in gold["cases"]:
        raw = bytes.fromhex("".join(c["macs"]))
        assert c_mac_crebuild(raw, c["n"], c["curve"], 0, c["write_step"]).hex() == "".join(c["X"])
        assert c_mac_crebuild(raw, c["n"], c["curve"], 1, c["write_step"]).hex() == "".join(c["Y"])


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_c_mac_mix_vs_python_and_vs_crebuild_last_stage(curve):
    """oracle_icc_mac_mix vs icc_py.mac_mix; and mix of the two half-size transforms == the full CRebuild network
    (the incremental construction the code is named for: Server.hpp:1209-1328 vs :1548-1687)"""
    import icc_py
    n = 8
    macs = [icc_py.ec_mul(curve, GX[curve], 555 + 31 * i) for i in range(n)]
    macs[5] = None
    half = n // 2
    a0, a1 = macs[:half], macs[half:]
    want = icc_py.mac_mix(a0, a1, n, curve)
    out = ctypes.create_string_buffer(64 * n)
    common.oracle().oracle_icc_mac_mix(b"".join(pt_bytes(p) for p in a0), b"".join(pt_bytes(p) for p in a1), ctypes.c_size_t(half),
                                       ctypes.c_size_t(n), 0 if curve == "bn254" else 1, out, 2)
    assert out.raw == b"".join(pt_bytes(p) for p in want)
