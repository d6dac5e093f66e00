"""Parity of the HIP ICC encode against the oracle (C restatement working directly in Z/LCM, itself pinned to the
Python restatement of porla/Server/Server.hpp:1487-1833 + :531-541), through the C ABI.  Bit-exact."""
import ctypes
import hashlib

import pytest

from tests import common

pytestmark = pytest.mark.gpu


def rows_bytes(n, ncols, seed=0):
    """uniform 256-bit chunks from a SHA-256 counter stream (SURVEY.md s8d cfg 5)"""
    out = bytearray()
    i = 0
    while len(out) < 32 * n * ncols:
        out += hashlib.sha256(b"porla-icc" + seed.to_bytes(4, "little") + i.to_bytes(8, "little")).digest()
        i += 1
    return bytes(out[:32 * n * ncols])


def oracle_crebuild(rows, n, ncols, curve, part, write_step):
    L = common.oracle()
    x = ctypes.create_string_buffer(64 * n * ncols)
    al = ctypes.create_string_buffer(32 * n * ncols)
    sc = ctypes.create_string_buffer(32 * n * ncols)
    L.oracle_icc_crebuild(rows, ctypes.c_size_t(n), ctypes.c_size_t(ncols), curve, part, ctypes.c_uint64(write_step), x, al, sc,
                          common.ncpu())
    return x.raw, al.raw, sc.raw


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n,ncols", [(2, 128), (4, 5), (8, 128), (16, 3), (32, 128), (1024, 128), (2048, 16)])
def test_crebuild_x_part(curve, n, ncols):
    from porla_amd import icc
    rows = rows_bytes(n, ncols, seed=n)
    got = icc.crebuild_host(rows, n, ncols, curve, 0, 0)
    want = oracle_crebuild(rows, n, ncols, icc.CURVE[curve], 0, 0)
    assert got[0] == want[0]      # values mod LCM, 64-byte LE
    assert got[1] == want[1]      # values mod p_icc
    assert got[2] == want[2]      # alignment scalars, big-endian


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_crebuild_y_part_and_scalar_formats(curve):
    """Y = X * wt with wt = w^reverse_bits(write_step % N, height-1) (Server.hpp:1494); at write_step % N == 0, Y == X"""
    from porla_amd import icc
    n, ncols = 64, 128
    rows = rows_bytes(n, ncols, seed=9)
    for ws in (0, 5, 37, 64 + 3):
        got = icc.crebuild_host(rows, n, ncols, curve, ws, 1)
        want = oracle_crebuild(rows, n, ncols, icc.CURVE[curve], 1, ws)
        assert got == want
    assert icc.crebuild_host(rows, n, ncols, curve, 0, 1)[0] == icc.crebuild_host(rows, n, ncols, curve, 0, 0)[0]
    # little-endian limb form of the scalars (secp256k1_scalar layout)
    be = icc.crebuild_host(rows, n, ncols, curve, 0, 0)[2]
    le = icc.crebuild_host(rows, n, ncols, curve, 0, 0, scalar_le=True)[2]
    assert all(be[32 * i:32 * i + 32] == le[32 * i:32 * i + 32][::-1] for i in range(n * ncols))


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_edge_values(curve):
    """chunks equal to 0, 2^256-1, p_icc, p_icc-1, q, q-1 (unreduced inputs, Client.hpp:369-371 allows any 256-bit value)"""
    from porla_amd import icc
    import icc_py
    n, ncols = 8, 6
    vals = [0, 2**256 - 1, icc_py.P_ICC, icc_py.P_ICC - 1, icc_py.Q[curve], icc_py.Q[curve] - 1, 1, 2**255]
    rows = b"".join(vals[(r + c) % len(vals)].to_bytes(32, "little") for r in range(n) for c in range(ncols))
    got = icc.crebuild_host(rows, n, ncols, curve, 3, 1)
    assert got == oracle_crebuild(rows, n, ncols, icc.CURVE[curve], 1, 3)


def test_full_size_2_22_elements():
    """BASELINE.json config 5: 2^22 elements = 2^15 rows x 128 columns, device-resident, vs the oracle"""
    import torch
    from porla_amd import icc
    n, ncols = 1 << 15, 128
    rows = rows_bytes(n, ncols, seed=22)
    d_in = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    d_x = torch.empty(64 * n * ncols, dtype=torch.uint8, device="cuda")
    d_al = torch.empty(32 * n * ncols, dtype=torch.uint8, device="cuda")
    d_sc = torch.empty(32 * n * ncols, dtype=torch.uint8, device="cuda")
    icc.crebuild_device(d_in.data_ptr(), n, ncols, "bn254", 0, 0, d_x.data_ptr(), d_al.data_ptr(), d_sc.data_ptr(),
                        stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = oracle_crebuild(rows, n, ncols, 0, 0, 0)
    assert d_x.cpu().numpy().tobytes() == want[0]
    assert d_al.cpu().numpy().tobytes() == want[1]
    assert d_sc.cpu().numpy().tobytes() == want[2]


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n,ncols,part,ws", [(1 << 17, 3, 0, 0), (1 << 18, 1, 1, 99991), (1 << 16, 5, 1, 7), (1 << 19, 1, 0, 12345), (1 << 19, 2, 1, 3)])
def test_three_passes_and_worst_case_operand_growth(curve, n, ncols, part, ws):
    """up to 9 stages per LDS-fused pass (round 5): 2^16 .. 2^18 rows are two long passes, 2^19 rows three (the middle-pass
    instantiation of the kernel); and the longest chains of
    unreduced butterflies of the reduced-radix kernel (icc30.hip.h: rows that meet only unit twiddles are brought back every fourth
    stage); inputs at the top of the range (every chunk 2^256 - 1 or p_icc - 1) make every sum as large as it can get"""
    from porla_amd import icc
    import icc_py
    big = [(1 << 256) - 1, icc_py.P_ICC - 1, icc_py.Q[curve] - 1]
    rows = b"".join(big[(r * 7 + c) % 3].to_bytes(32, "little") for r in range(n) for c in range(ncols))
    got = icc.crebuild_host(rows, n, ncols, curve, ws, part)
    assert got == oracle_crebuild(rows, n, ncols, icc.CURVE[curve], part, ws)
    rows = rows_bytes(n, ncols, seed=n + ws)
    got = icc.crebuild_host(rows, n, ncols, curve, ws, part, want_x=False)
    want = oracle_crebuild(rows, n, ncols, icc.CURVE[curve], part, ws)
    assert got[1] == want[1] and got[2] == want[2]


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n,ncols,ws", [(2, 128, 1), (16, 3, 5), (256, 128, 77), (2048, 16, 1234567), (1 << 17, 3, 99991)])
def test_both_parts_from_one_network(curve, n, ncols, ws):
    """porla_icc_encode_xy_device: the network is linear over Z/LCM and the Y part is the X part's network on chunks scaled by wt
    (Server.hpp:1494, :1512-1522, :1691-1830), so Y_k = wt X_k mod LCM -- one product per residue and symbol in the last pass
    instead of a second encode.  All six outputs against the oracle's two separate encodes (1, 2 and 3 passes; edge-valued chunks
    among the rows), then every output on its own (each subset takes its own way through the finish step)."""
    import torch
    from porla_amd import icc
    import icc_py
    rows = bytearray(rows_bytes(n, ncols, seed=n + ws))
    edge = [0, 2**256 - 1, icc_py.P_ICC, icc_py.P_ICC - 1, icc_py.Q[curve], icc_py.Q[curve] - 1]
    for k, v in enumerate(edge):
        if k < n * ncols:
            rows[32 * k:32 * k + 32] = v.to_bytes(32, "little")
    rows = bytes(rows)
    want = [oracle_crebuild(rows, n, ncols, icc.CURVE[curve], part, ws) for part in (0, 1)]
    d_in = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    widths = (64, 32, 32)
    outs = [[torch.empty(w * n * ncols, dtype=torch.uint8, device="cuda") for w in widths] for _ in (0, 1)]
    icc.crebuild_xy_device(d_in.data_ptr(), n, ncols, curve, ws, *[t.data_ptr() for t in outs[0]], *[t.data_ptr() for t in outs[1]],
                           stream=stream)
    torch.cuda.synchronize()
    for part in (0, 1):
        for k, name in enumerate(("values mod LCM", "values mod p_icc", "alignment scalars")):
            assert bytes(outs[part][k].cpu().numpy()) == want[part][k], "%s, part %d" % (name, part)
    if n <= 2048:
        names = ("d_x", "d_aligned", "d_scalars", "d_y_x", "d_y_aligned", "d_y_scalars")
        for k, name in enumerate(names):
            t = torch.zeros(widths[k % 3] * n * ncols, dtype=torch.uint8, device="cuda")
            icc.crebuild_xy_device(d_in.data_ptr(), n, ncols, curve, ws, stream=stream, **{name: t.data_ptr()})
            torch.cuda.synchronize()
            assert bytes(t.cpu().numpy()) == want[k // 3][k % 3], name
