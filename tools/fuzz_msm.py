#!/usr/bin/env python3
"""Randomised differential test of the MSM against the oracle: sizes, window overrides, GLV on/off, scalar distributions
(uniform 256-bit, small ints, few distinct values, near the group order / lambda), repeated and infinity points, both curves."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from porla_amd import multiexp as mx, lib
from tests import common
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
NMAX = 40000          # spans the single-launch path (<= 32768 pairs) and the general one
pools = {"bn254": common.synth_points(NMAX), "secp256k1": common.secp_bench_points(NMAX)}
ORD = {"bn254": 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001, "secp256k1": common.SECP_N}
LAM = {"bn254": 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23,
       "secp256k1": 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72}
t_end = time.time() + seconds
t_note = time.time() + 60          # a progress line per minute: a silent GPU command is taken to be hung
cases = fails = 0
while time.time() < t_end:
    if time.time() > t_note:
        print("... %d cases so far" % cases, flush=True)
        t_note = time.time() + 60
    curve = rnd.choice(["bn254", "secp256k1"])
    n = rnd.choice([1, 2, 3, 7, 64, 65, 127, 128, 129, 500, 1408, 3200, rnd.randrange(1, NMAX)])
    dist = rnd.choice(["uniform", "small", "fewvals", "edge", "bits"])
    q = ORD[curve]
    if dist == "uniform":
        vals = [rnd.getrandbits(256) for _ in range(n)]
    elif dist == "small":
        b = rnd.choice([1, 8, 16, 31, 32, 33, 64])
        vals = [rnd.getrandbits(b) for _ in range(n)]
    elif dist == "fewvals":
        base = [rnd.getrandbits(256) for _ in range(rnd.choice([1, 2, 5]))]
        vals = [rnd.choice(base) for _ in range(n)]
    elif dist == "bits":
        b = rnd.randrange(1, 257)
        vals = [rnd.getrandbits(b) | (1 << (b - 1)) for _ in range(n)]
    else:
        e = [0, 1, q - 1, q, q + 1, 2 * q + 3, (1 << 256) - 1, LAM[curve], q - LAM[curve], LAM[curve] + 1, (q - 1) // 2, 1 << 128, (1 << 128) - 1, 1 << 127]
        vals = [rnd.choice(e) for _ in range(n)]
    sc = b"".join((v & ((1 << 256) - 1)).to_bytes(32, "big") for v in vals)
    pool = pools[curve]
    pmode = rnd.choice(["distinct", "repeat", "withinf", "pairs"])
    if pmode == "distinct":
        off = rnd.randrange(0, NMAX - n + 1)
        pt = pool[64 * off:64 * (off + n)]
    elif pmode == "repeat":
        k = rnd.choice([1, 2, 7, 50])
        pt = b"".join(pool[64 * (i % k):64 * (i % k) + 64] for i in range(n))
    elif pmode == "withinf":
        pt = b"".join(bytes(64) if rnd.random() < 0.2 else pool[64 * i:64 * i + 64] for i in range(n))
    else:   # P, -P pairs
        P = (1 << 256) - (1 << 32) - 977 if curve == "secp256k1" else 21888242871839275222246405745257275088696311157297823662689037894645226208583
        out = []
        for i in range(n):
            p = pool[64 * (i // 2):64 * (i // 2) + 64]
            if i & 1:
                p = p[:32] + (P - int.from_bytes(p[32:], "big")).to_bytes(32, "big")
            out.append(p)
        pt = b"".join(out)
    c = rnd.choice([0, 0, 0, rnd.randrange(2, 21)])
    glv = rnd.choice([-1, 0, 1])
    # the single-launch path for n <= 4096 (automatic or forced window width), or switched off; sometimes the range-sharded entry
    small_on, small_c = rnd.choice([(1, 0), (1, 0), (1, rnd.randrange(1, 9)), (0, 0)])
    shards = rnd.choice([0, 0, 0, 0, rnd.randrange(1, 7)])
    lib.porla_gpu_set_msm_window(c)
    lib.porla_gpu_set_msm_glv(glv)
    lib.porla_gpu_set_msm_small(small_on, small_c)
    got = mx.msm_host_multi(curve, sc, pt, n, shards=shards, devices=1) if shards else mx.msm_host(curve, sc, pt, n)
    want = common.oracle_msm(sc, pt, n) if curve == "bn254" else common.oracle_secp_msm(sc, pt, n)
    cases += 1
    if got != want:
        fails += 1
        print("MISMATCH", curve, "n=%d" % n, dist, pmode, "c=%d" % c, "glv=%d" % glv, "small=%d/%d" % (small_on, small_c), "shards=%d" % shards, flush=True)
lib.porla_gpu_set_msm_small(1, 0)
lib.porla_gpu_set_msm_window(0)
lib.porla_gpu_set_msm_glv(-1)
print("fuzz: %d cases, %d mismatches (seed %d)" % (cases, fails, seed))
sys.exit(1 if fails else 0)
