"""Parity of the HIP audit row combine (porla_amd/csrc/audit.hip) against the Python restatement of Server::audit's
B += coeff * row loop (porla/Server/Server.hpp:790-828) followed by the scalar part of align_MAC (Server.hpp:531-541).
Bit-exact on the exact sums, the rows mod p_icc and the alignment scalars."""
import random

import pytest

pytestmark = pytest.mark.gpu


def run(curve, n64, n32, n_cols, seed, store64=37, store32=29, edge=False):
    import numpy as np
    import torch
    import icc_py
    from porla_amd import icc
    rnd = random.Random(seed)
    lcm, p = icc_py.LCM[curve], icc_py.P_ICC
    s64 = [[rnd.randrange(lcm) for _ in range(n_cols)] for _ in range(store64)]
    s32 = [[rnd.randrange(p) for _ in range(n_cols)] for _ in range(store32)]
    if edge:
        s64[0] = [lcm - 1] * n_cols
        s64[1] = [0] * n_cols
        s32[0] = [p - 1] * n_cols
    idx64 = [rnd.randrange(store64) for _ in range(n64)]
    idx32 = [rnd.randrange(store32) for _ in range(n32)]
    c64 = [rnd.getrandbits(31) for _ in range(n64)]
    c32 = [rnd.getrandbits(31) for _ in range(n32)]
    if edge and n64 > 2:
        idx64[0], c64[0] = 0, 0x7fffffff
        idx64[1], c64[1] = 0, 0xffffffff          # abs(INT_MIN) read back as unsigned
        idx64[2], c64[2] = 1, 12345
    B, mods, cs = icc_py.audit_combine([s64[i] for i in idx64] + [s32[i] for i in idx32], c64 + c32, curve)
    d_s64 = torch.frombuffer(bytearray(b"".join(v.to_bytes(64, "little") for r in s64 for v in r)), dtype=torch.uint8).cuda()
    d_s32 = torch.frombuffer(bytearray(b"".join(v.to_bytes(32, "little") for r in s32 for v in r)), dtype=torch.uint8).cuda()
    d_i64 = torch.tensor(idx64 or [0], dtype=torch.int64).cuda()
    d_i32 = torch.tensor(idx32 or [0], dtype=torch.int64).cuda()
    d_c64 = torch.tensor(np.array(c64 or [0], dtype=np.uint32).view(np.int32)).cuda()
    d_c32 = torch.tensor(np.array(c32 or [0], dtype=np.uint32).view(np.int32)).cuda()
    d_ex = torch.empty(80 * n_cols, dtype=torch.uint8, device="cuda")
    d_al = torch.empty(32 * n_cols, dtype=torch.uint8, device="cuda")
    d_be = torch.empty(32 * n_cols, dtype=torch.uint8, device="cuda")
    d_sc = torch.empty(32 * n_cols, dtype=torch.uint8, device="cuda")
    icc.audit_combine_device(d_s64.data_ptr(), d_i64.data_ptr(), d_c64.data_ptr(), n64, d_s32.data_ptr(), d_i32.data_ptr(),
                             d_c32.data_ptr(), n32, n_cols, curve, d_ex.data_ptr(), d_al.data_ptr(), d_be.data_ptr(),
                             d_sc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ex, al, be, sc = (bytes(t.cpu().numpy()) for t in (d_ex, d_al, d_be, d_sc))
    for j in range(n_cols):
        assert int.from_bytes(ex[80 * j:80 * j + 80], "little") == B[j]
        assert int.from_bytes(al[32 * j:32 * j + 32], "little") == mods[j]
        assert int.from_bytes(be[32 * j:32 * j + 32], "big") == mods[j]
        assert int.from_bytes(sc[32 * j:32 * j + 32], "big") == cs[j]


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
@pytest.mark.parametrize("n64,n32", [(1, 0), (0, 1), (5, 3), (1408, 0), (1280, 1920)])
def test_matches_python_restatement(curve, n64, n32):
    """1 408 = NUM_CHECK_AUDIT * height at N = 2^10; 3 200 at N = 2^24 with the top levels as 256-bit rows"""
    run(curve, n64, n32, 128, seed=n64 * 7 + n32)


def test_edge_values_and_ragged_columns():
    run("bn254", 40, 9, 128, seed=1, edge=True)
    run("secp256k1", 40, 9, 5, seed=2, edge=True)       # fewer columns than a wave
    run("bn254", 300, 0, 200, seed=3)                    # more than 128 columns: second column block


def test_large_challenge_takes_the_eight_slice_kernel():
    """more than 16 384 challenged rows: k_audit_accumulate<8> (the bandwidth-regime instantiation), both row formats, ragged split"""
    run("bn254", 12001, 6007, 128, seed=99, edge=True)
    run("secp256k1", 17000, 0, 128, seed=100)
