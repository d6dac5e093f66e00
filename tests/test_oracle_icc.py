"""CPU: pins the data-side ICC oracle.  (1) the Python restatement (oracle/icc_py.py: loop-for-loop CRebuild_Cached,
porla/Server/Server.hpp:1487-1833, and align_MAC's scalars, :531-541) against the committed fixture
tests/golden/icc_golden.json, stage by stage; (2) the C oracle (oracle/icc_ref.c, which works in Z/LCM with its own 8 x 64-bit
Montgomery multiplier) against the same fixture and against Python on larger shapes; (3) the X part mod p_icc against an
independent recursive formulation of the butterfly network.  The reference holds no tests for this path and NTL is absent
(SURVEY.md s8c): parity w.r.t. NTL is unpinned, the fixture pins the restatement across implementations."""
import ctypes
import hashlib
import json
import os

import pytest

from tests import common

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "icc_golden.json")))
CURVE_ID = {"bn254": 0, "secp256k1": 1}


def le(h):
    return int.from_bytes(bytes.fromhex(h), "little")


def c_crebuild(rows_bytes, n, ncols, curve, is_y, write_step):
    L = common.oracle()
    x = ctypes.create_string_buffer(64 * n * ncols)
    al = ctypes.create_string_buffer(32 * n * ncols)
    sc = ctypes.create_string_buffer(32 * n * ncols)
    L.oracle_icc_crebuild(rows_bytes, ctypes.c_size_t(n), ctypes.c_size_t(ncols), CURVE_ID[curve], is_y, ctypes.c_uint64(write_step),
                          x, al, sc, 2)
    return x.raw, al.raw, sc.raw


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "%s-%d" % (c["curve"], c["n"]))
def test_python_restatement_reproduces_the_fixture(case):
    import icc_py
    rows = [[le(v) for v in r] for r in case["rows"]]
    trace = []
    X, Y = icc_py.crebuild(rows, case["curve"], case["write_step"], trace)
    stages = trace if case["n"] <= 16 else trace[-1:]
    assert [[[le(v) for v in r] for r in st] for st in case["X_after_stage"]] == stages
    assert [[le(v) for v in r] for r in case["X"]] == X and [[le(v) for v in r] for r in case["Y"]] == Y
    for r, mods, cs in zip(X, case["X_mod_p_icc"], case["X_align_scalars"]):
        m, c = icc_py.align(r, case["curve"])
        assert m == [le(v) for v in mods] and c == [le(v) for v in cs]


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "%s-%d" % (c["curve"], c["n"]))
def test_c_oracle_reproduces_the_fixture(case):
    n, ncols = case["n"], case["ncols"]
    rows_bytes = b"".join(bytes.fromhex(v) for r in case["rows"] for v in r)
    for is_y, key in ((0, "X"), (1, "Y")):
        x, al, sc = c_crebuild(rows_bytes, n, ncols, case["curve"], is_y, case["write_step"])
        want = b"".join(bytes.fromhex(v) for r in case[key] for v in r)
        assert x == want
        if key == "X":
            assert al == b"".join(bytes.fromhex(v) for r in case["X_mod_p_icc"] for v in r)
            # the C oracle emits the alignment scalars big-endian (the bn254_scalar / secp256k1 scalar wire order)
            assert sc == b"".join(bytes.fromhex(v)[::-1] for r in case["X_align_scalars"] for v in r)


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n,ncols,ws", [(2, 3, 0), (4, 128, 3), (128, 5, 77), (256, 2, 255)])
def test_c_oracle_vs_python_on_other_shapes(curve, n, ncols, ws):
    import icc_py
    vals = []
    i = 0
    while len(vals) < n * ncols:
        vals.append(int.from_bytes(hashlib.sha256(b"icc-shape" + i.to_bytes(8, "little")).digest(), "little"))
        i += 1
    rows = [vals[r * ncols:(r + 1) * ncols] for r in range(n)]
    rows_bytes = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    X, Y = icc_py.crebuild(rows, curve, ws)
    for is_y, part in ((0, X), (1, Y)):
        x, al, sc = c_crebuild(rows_bytes, n, ncols, curve, is_y, ws)
        assert x == b"".join(v.to_bytes(64, "little") for r in part for v in r)
        assert al == b"".join((v % icc_py.P_ICC).to_bytes(32, "little") for r in part for v in r)
        q = icc_py.Q[curve]
        assert sc == b"".join((((v % icc_py.P_ICC) - v) % q).to_bytes(32, "big") for r in part for v in r)


@pytest.mark.parametrize("n", [2, 8, 64, 512])
def test_x_part_mod_p_icc_equals_an_independent_recursive_formulation(n):
    import icc_py
    col = [int.from_bytes(hashlib.sha256(b"icc-net" + i.to_bytes(4, "little")).digest(), "little") for i in range(n)]
    X, _ = icc_py.crebuild([[v] for v in col], "bn254", 0)
    want = icc_py.linear_network_matrix(n)([v % icc_py.P_ICC for v in col])
    assert [r[0] % icc_py.P_ICC for r in X] == want


def test_hadd_and_hrebuild_restatements_are_consistent_with_the_encode():
    """oracle/icc_py.py:hadd / hrebuild (Server::HAdd, HRebuildX: Server.hpp:1388-1477, 1329-1386) against the restatement of
    CRebuild they must agree with: adding N blocks one by one with HAdd / HRebuild builds, level by level, what the batch encode of
    those N blocks gives mod p_icc (the hierarchical log structure's invariant), for the X part"""
    import icc_py
    import random
    rnd = random.Random(3)
    N, ncols = 8, 3
    blocks = [[rnd.getrandbits(256) for _ in range(ncols)] for _ in range(N)]
    # HAdd with write_step = t: data_B2 = data * wt aligned, c = (mod - A) % q, MAC_B2 = wt * MAC
    for t in (0, 1, 5):
        mods, cs, mac_b2, wt = icc_py.hadd(blocks[t], None, N, t, "bn254")
        assert wt == pow(icc_py.root_w(N), icc_py.reverse_bits(t % N, icc_py.height_of(N) - 1), icc_py.P_ICC)
        for d, m, c in zip(blocks[t], mods, cs):
            assert m == d * wt % icc_py.P_ICC and (d * wt + c) % icc_py.Q["bn254"] == m % icc_py.Q["bn254"] and mac_b2 is None
    # X part: level L after 2^L insertions == mix tree of those blocks; compare with mixing by hand
    levels = [[[0] * ncols for _ in range(2 << i)] for i in range(4)]
    levels[0][0], levels[0][1] = blocks[0], blocks[1]
    icc_py.hrebuild(levels, 1, N, "bn254")
    want1 = icc_py.mix([blocks[0]], [blocks[1]], N, "bn254")
    assert levels[1][:2] == want1 and levels[1][2:4] == want1
    # next pair arrives: level 0 again, then a level-2 rebuild mixes level 1's resident half with the new pair's mix
    levels[0][0], levels[0][1] = blocks[2], blocks[3]
    icc_py.hrebuild(levels, 2, N, "bn254")
    want2 = icc_py.mix(want1, icc_py.mix([blocks[2]], [blocks[3]], N, "bn254"), N, "bn254")
    assert levels[2][:4] == want2
