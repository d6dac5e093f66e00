#!/usr/bin/env python3
"""Randomised differential test of the ICC kernels against the oracle (oracle/icc_ref.c, oracle/mac_ref.c): data encode
(CRebuild: sizes 2 .. 2^13 rows, 1 .. 128 columns, X / Y part, any write_step, uniform and edge-valued chunks), the incremental
mix, the MAC-side encode and mix (points incl. infinity), both curves.  usage: fuzz_icc.py [seconds] [seed]"""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from porla_amd import icc
from tests import common
import icc_py
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
L = common.oracle()
NCPU = common.ncpu()
MAC_POOL = 512
mac_pool = {"bn254": common.synth_points(MAC_POOL, start=7000)}
sc = common.secp_bench_scalars(MAC_POOL, start=900)
buf = ctypes.create_string_buffer(64 * MAC_POOL)
L.oracle_secp256k1_mul_g_batch(sc, ctypes.c_size_t(MAC_POOL), buf, NCPU)
mac_pool["secp256k1"] = buf.raw


def chunks(n, curve, width):
    """n values below 2^(8 width) as `width`-byte little-endian words; uniform, or drawn from the edge set"""
    mode = rnd.choice(["uniform", "uniform", "edge", "small"])
    if mode == "uniform":
        return rnd.randbytes(width * n)
    top = icc_py.LCM[curve] if width == 64 else 1 << 256
    if mode == "small":
        return b"".join(rnd.randrange(1 << 16).to_bytes(width, "little") for _ in range(n))
    e = [0, 1, top - 1, icc_py.P_ICC, icc_py.P_ICC - 1, icc_py.Q[curve], icc_py.Q[curve] - 1, 1 << 255, (1 << 248) - 1]
    e = [v % top for v in e]
    return b"".join(rnd.choice(e).to_bytes(width, "little") for _ in range(n))


def lcm_values(n, curve):
    lcm = icc_py.LCM[curve]
    if rnd.random() < 0.3:
        e = [0, 1, lcm - 1, icc_py.P_ICC, icc_py.Q[curve], lcm - icc_py.P_ICC]
        return b"".join(rnd.choice(e).to_bytes(64, "little") for _ in range(n))
    return b"".join(rnd.randrange(lcm).to_bytes(64, "little") for _ in range(n))


def macs(n, curve):
    out = []
    for _ in range(n):
        if rnd.random() < 0.1:
            out.append(bytes(64))
        else:
            j = rnd.randrange(MAC_POOL)
            out.append(mac_pool[curve][64 * j:64 * j + 64])
    return b"".join(out)


t_end = time.time() + seconds
t_note = time.time() + 60
cases = fails = 0
kinds = {}
while time.time() < t_end:
    if time.time() > t_note:
        print("... %d cases so far" % cases, flush=True)
        t_note = time.time() + 60
    curve = rnd.choice(["bn254", "secp256k1"])
    cid = icc.CURVE[curve]
    kind = rnd.choice(["encode", "encode", "mix", "mac_encode", "mac_mix"])
    desc = ""
    if kind == "encode":
        n = 1 << rnd.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 11, 12, 13])
        ncols = rnd.choice([1, 2, 3, 16, 127, 128, rnd.randrange(1, 129)])
        if n * ncols > 1 << 19:
            ncols = max(1, (1 << 19) // n)
        part, ws = rnd.choice([0, 1]), rnd.choice([0, 1, n - 1, n, rnd.randrange(1 << 40)])
        rows = chunks(n * ncols, curve, 32)
        got = icc.crebuild_host(rows, n, ncols, curve, ws, part)
        x = ctypes.create_string_buffer(64 * n * ncols)
        al = ctypes.create_string_buffer(32 * n * ncols)
        s = ctypes.create_string_buffer(32 * n * ncols)
        L.oracle_icc_crebuild(rows, ctypes.c_size_t(n), ctypes.c_size_t(ncols), cid, part, ctypes.c_uint64(ws), x, al, s, NCPU)
        ok = tuple(got) == (x.raw, al.raw, s.raw)
        desc = "n=%d ncols=%d part=%d ws=%d" % (n, ncols, part, ws)
    elif kind == "mix":
        length = 1 << rnd.choice([0, 1, 2, 3, 5, 8, 10, 12])
        ncols = rnd.choice([1, 5, 128, rnd.randrange(1, 129)])
        if length * ncols > 1 << 16:
            ncols = max(1, (1 << 16) // length)
        n_total = (2 * length) << rnd.choice([0, 0, 1, 3])
        a0, a1 = lcm_values(length * ncols, curve), lcm_values(length * ncols, curve)
        got = icc.mix_host(a0, a1, length, ncols, n_total, curve)
        want = ctypes.create_string_buffer(2 * length * ncols * 64)
        L.oracle_icc_mix(a0, a1, ctypes.c_size_t(length), ctypes.c_size_t(ncols), ctypes.c_size_t(n_total), cid, want)
        ok = got == want.raw
        desc = "len=%d ncols=%d n_total=%d" % (length, ncols, n_total)
    elif kind == "mac_encode":
        n = 1 << rnd.choice([1, 2, 3, 4, 5, 6, 7, 8])
        part, ws = rnd.choice([0, 1]), rnd.choice([0, 1, n - 1, rnd.randrange(1 << 40)])
        m = macs(n, curve)
        want = ctypes.create_string_buffer(64 * n)
        if rnd.random() < 0.4:
            # both halves from one network (porla_icc_mac_encode_xy_host: Y_k = wt * X_k) against the oracle's two encodes; the
            # matrix form too at these small sizes every fourth time (the ladder is the default at every size)
            from porla_amd import lib
            matrix = rnd.random() < 0.25
            if matrix:
                lib.porla_icc_mac_set_matrix_max(512)
            gx, gy = icc.mac_crebuild_xy_host(m, n, curve, ws)
            if matrix:
                lib.porla_icc_mac_set_matrix_max(0)
            ok = True
            for p_, g_ in ((0, gx), (1, gy)):
                L.oracle_icc_mac_crebuild(m, ctypes.c_size_t(n), cid, p_, ctypes.c_uint64(ws), want, NCPU)
                ok = ok and g_ == want.raw
            desc = "xy n=%d ws=%d matrix=%s" % (n, ws, matrix)
        else:
            got = icc.mac_crebuild_host(m, n, curve, ws, part)
            L.oracle_icc_mac_crebuild(m, ctypes.c_size_t(n), cid, part, ctypes.c_uint64(ws), want, NCPU)
            ok = got == want.raw
            desc = "n=%d part=%d ws=%d" % (n, part, ws)
    else:
        length = 1 << rnd.choice([0, 1, 2, 4, 6, 8])
        n_total = (2 * length) << rnd.choice([0, 0, 1, 4])
        a0, a1 = macs(length, curve), macs(length, curve)
        got = icc.mac_mix_host(a0, a1, length, n_total, curve)
        want = ctypes.create_string_buffer(2 * length * 64)
        L.oracle_icc_mac_mix(a0, a1, ctypes.c_size_t(length), ctypes.c_size_t(n_total), cid, want, NCPU)
        ok = got == want.raw
        desc = "len=%d n_total=%d" % (length, n_total)
    cases += 1
    kinds[kind] = kinds.get(kind, 0) + 1
    if not ok:
        fails += 1
        print("MISMATCH", kind, curve, desc, flush=True)
print("fuzz icc: %d cases %s, %d mismatches (seed %d)" % (cases, kinds, fails, seed))
sys.exit(1 if fails else 0)
