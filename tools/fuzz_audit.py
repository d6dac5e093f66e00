#!/usr/bin/env python3
"""Randomised differential test of the audit-side entry points against the oracle: the row combine (porla_audit_combine_device:
1 .. 20 000 challenged rows of both row formats, 1 .. 200 columns, edge-valued symbols and coefficients; oracle/icc_py.py), the
gathered pair of MSMs (porla_*_audit_msm_pair_device: repeated indices, infinity entries, a point and its negative; oracle MSMs on the
host-gathered arrays, both curves) and the digest batch (porla_kzg_digest_batch_device: coefficients >= r, zero rows; against the
one-row symbol compute_digest, itself pinned to the oracle by tests/test_fixed_base_gpu.py; the MAC batch on the same rows against
add_point(digest, complement), the complements against mult_point(h_MAC, scalar)).  usage: fuzz_audit.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from porla_amd import icc, multiexp as mx
from tests import common
import icc_py
import bn254_py
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
POOL = 256
pool = {"bn254": common.synth_points(POOL, start=11000), "secp256k1": common.secp_bench_points(POOL)}
neg0 = bn254_py.neg_point(pool["bn254"][:64])
mx.init_key(bytes(range(1, 17)), bytes(range(33, 49)))
mx.init_SRS(128)
stream = torch.cuda.current_stream().cuda_stream
dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()


def combine_case():
    curve = rnd.choice(["bn254", "secp256k1"])
    lcm, p = icc_py.LCM[curve], icc_py.P_ICC
    n_cols = rnd.choice([1, 5, 64, 128, 128, 128, 200])
    big = rnd.random() < 0.15
    total = rnd.randrange(16385, 20001) if big else rnd.choice([1, 2, 3, 7, 31, 128, 1408, 3200, rnd.randrange(1, 4000)])
    if big:
        n_cols = min(n_cols, 16)
    n64 = rnd.choice([0, total, rnd.randrange(total + 1)])
    n32 = total - n64
    st64, st32 = rnd.randrange(1, 40), rnd.randrange(1, 40)
    edge64 = [0, 1, lcm - 1, p, p - 1, lcm - p]
    edge32 = [0, 1, p - 1, 1 << 255, (1 << 248) - 1]
    sym = lambda top, edge: rnd.choice(edge) if rnd.random() < 0.2 else rnd.randrange(top)
    s64 = [[sym(lcm, edge64) for _ in range(n_cols)] for _ in range(st64)]
    s32 = [[sym(p, edge32) for _ in range(n_cols)] for _ in range(st32)]
    cf = lambda: rnd.choice([0, 1, 0x7fffffff, 0x80000000, 0xffffffff]) if rnd.random() < 0.1 else rnd.getrandbits(31)
    i64, i32 = [rnd.randrange(st64) for _ in range(n64)], [rnd.randrange(st32) for _ in range(n32)]
    c64, c32 = [cf() for _ in range(n64)], [cf() for _ in range(n32)]
    B, mods, cs = icc_py.audit_combine([s64[i] for i in i64] + [s32[i] for i in i32], c64 + c32, curve)
    d = [dev(b"".join(v.to_bytes(64, "little") for r in s64 for v in r)), dev(b"".join(v.to_bytes(32, "little") for r in s32 for v in r)),
         torch.tensor(i64 or [0], dtype=torch.int64).cuda(), torch.tensor(i32 or [0], dtype=torch.int64).cuda(),
         torch.tensor(np.array(c64 or [0], dtype=np.uint32).view(np.int32)).cuda(), torch.tensor(np.array(c32 or [0], dtype=np.uint32).view(np.int32)).cuda()]
    o = [torch.empty(k * n_cols, dtype=torch.uint8, device="cuda") for k in (80, 32, 32, 32)]
    icc.audit_combine_device(d[0].data_ptr(), d[2].data_ptr(), d[4].data_ptr(), n64, d[1].data_ptr(), d[3].data_ptr(), d[5].data_ptr(), n32,
                             n_cols, curve, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), stream)
    torch.cuda.synchronize()
    ex, al, be, sc = (bytes(t.cpu().numpy()) for t in o)
    ok = all(int.from_bytes(ex[80 * j:80 * j + 80], "little") == B[j] and int.from_bytes(al[32 * j:32 * j + 32], "little") == mods[j]
             and int.from_bytes(be[32 * j:32 * j + 32], "big") == mods[j] and int.from_bytes(sc[32 * j:32 * j + 32], "big") == cs[j]
             for j in range(n_cols))
    return ok, ("combine", curve, n64, n32, n_cols)


def pair_case():
    curve = rnd.choice(["bn254", "secp256k1"])
    store = rnd.randrange(2, POOL)
    a = bytearray(pool[curve][:64 * store])
    if rnd.random() < 0.5:
        k = rnd.randrange(store)
        a[64 * k:64 * k + 64] = bytes(64)
    if curve == "bn254" and store > 3:
        a[64:128] = neg0                                    # entry 1 = -(entry 0)
    a = bytes(a)
    sh = rnd.randrange(store)
    b = a[64 * sh:] + a[:64 * sh]
    n = rnd.choice([1, 2, 3, 17, 128, 1408, 3200, rnd.randrange(1, 5000)])
    idx = [rnd.randrange(store) for _ in range(n)]
    coef = [rnd.choice([0, 1, 0x7fffffff, 0x80000000]) if rnd.random() < 0.1 else rnd.getrandbits(31) for _ in range(n)]
    if n > 4 and rnd.random() < 0.5:
        idx[0], idx[1], coef[0], coef[1] = 0, 1, 12345, 12345      # P and -P with the same coefficient
    d_a, d_b = dev(a), dev(b)
    d_i = torch.tensor(idx, dtype=torch.int64).cuda()
    d_c = torch.tensor(np.array(coef, dtype=np.uint32).view(np.int32)).cuda()
    torch.cuda.synchronize()
    if rnd.random() < 0.5:
        got = mx.audit_msm_pair_device(curve, d_a.data_ptr(), d_b.data_ptr(), d_i.data_ptr(), d_c.data_ptr(), n, stream)
    else:
        mx.audit_msm_pair_begin(rnd.randrange(1, 4), curve, d_a.data_ptr(), d_b.data_ptr(), d_i.data_ptr(), d_c.data_ptr(), n, stream)
        got = None
        for s in (1, 2, 3):
            try:
                got = mx.audit_msm_pair_end(s, curve)
                break
            except RuntimeError:
                pass
    sc = b"".join(c.to_bytes(32, "big") for c in coef)
    orc = common.oracle_msm if curve == "bn254" else common.oracle_secp_msm
    want = (orc(sc, b"".join(a[64 * i:64 * i + 64] for i in idx), n), orc(sc, b"".join(b[64 * i:64 * i + 64] for i in idx), n))
    return got == want, ("pair", curve, n, store)


def digest_case():
    n = rnd.choice([1, 2, 7, 8, 9, 63, 300, rnd.randrange(1, 2000)])
    rows = bytearray(rnd.randbytes(4096 * n))
    for _ in range(rnd.randrange(0, 6)):
        r, c = rnd.randrange(n), rnd.randrange(128)
        v = rnd.choice([0, 1, R - 1, R, R + 1, (1 << 256) - 1, 5 * R + 7])
        rows[4096 * r + 32 * c:4096 * r + 32 * c + 32] = v.to_bytes(32, "big")
    if rnd.random() < 0.3:
        r = rnd.randrange(n)
        rows[4096 * r:4096 * r + 4096] = bytes(4096)
    rows = bytes(rows)
    d_rows, d_out = dev(rows), torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    mx.kzg_digest_batch_device(d_rows.data_ptr(), n, d_out.data_ptr(), stream)
    torch.cuda.synchronize()
    got = bytes(d_out.cpu().numpy())
    check = sorted(set([0, n - 1] + [rnd.randrange(n) for _ in range(4)]))
    ok = all(got[64 * r:64 * r + 64] == mx.compute_digest(rows[4096 * r:4096 * r + 4096]) for r in check)
    # the MAC batch on the same rows: digest + complement(scalar), scalars with the edge values too
    sc = bytearray(b"".join(bytes(16) + rnd.randbytes(16) for _ in range(n)))
    for _ in range(rnd.randrange(0, 4)):
        r = rnd.randrange(n)
        sc[32 * r:32 * r + 32] = rnd.choice([0, 1, R - 1, R, R + 1, (1 << 256) - 1]).to_bytes(32, "big")
    d_sc, d_mac = dev(bytes(sc)), torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    mx.kzg_complement_batch_device(d_sc.data_ptr(), n, d_out.data_ptr(), stream)
    mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), n, d_mac.data_ptr(), stream)
    torch.cuda.synchronize()
    comp, macs = bytes(d_out.cpu().numpy()), bytes(d_mac.cpu().numpy())
    ok = ok and all(macs[64 * r:64 * r + 64] == mx.bn254_add(got[64 * r:64 * r + 64], comp[64 * r:64 * r + 64]) for r in range(n))
    ok = ok and all(comp[64 * r:64 * r + 64] == mx.bn254_mult(mx.compute_digest_complement((1).to_bytes(16, "big")), bytes(sc[32 * r:32 * r + 32]))
                    for r in check)
    return ok, ("digest", n)


t_end = time.time() + seconds
t_note = time.time() + 60
cases = fails = 0
kinds = {}
while time.time() < t_end:
    fn = rnd.choice([combine_case, pair_case, pair_case, digest_case])
    ok, what = fn()
    cases += 1
    kinds[what[0]] = kinds.get(what[0], 0) + 1
    if not ok:
        fails += 1
        print("MISMATCH", what, flush=True)
    if time.time() > t_note:
        print("... %d cases, %d mismatches %s" % (cases, fails, kinds), flush=True)
        t_note = time.time() + 60
print("fuzz_audit: %d cases %s, %d mismatches (seed %d)" % (cases, kinds, fails, seed))
sys.exit(1 if fails else 0)
