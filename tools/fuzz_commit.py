#!/usr/bin/env python3
"""Randomised differential test of the batched fixed-base commitments against the oracle: base size, window, row count,
coefficient count, row stride, both curves."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from porla_amd import multiexp as mx
from tests import common
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pools = {"bn254": common.synth_points(256), "secp256k1": common.secp_bench_points(256)}
t_end = time.time() + seconds
t_note = time.time() + 60          # a progress line per minute: a silent GPU command is taken to be hung
cases = fails = 0
while time.time() < t_end:
    if time.time() > t_note:
        print("... %d cases so far" % cases, flush=True)
        t_note = time.time() + 60
    curve = rnd.choice(["bn254", "secp256k1"])
    nb = rnd.choice([1, 2, 3, 17, 64, 127, 128, 200])
    c = rnd.choice([2, 3, 5, 8, 9, 12, 13])
    base = bytearray(pools[curve][:64 * nb])
    if rnd.random() < 0.2:
        k = rnd.randrange(nb)
        base[64 * k:64 * k + 64] = bytes(64)
    fb = mx.FixedBase(curve, bytes(base), nb, window_bits=c)
    for _ in range(4):
        n_rows = rnd.choice([1, 2, 5, 63, 64, 65, 300, rnd.randrange(1, 3000)])
        n_coeffs = rnd.choice([nb, max(1, nb - 1), rnd.randrange(1, nb + 1)])
        stride = 32 * n_coeffs + 32 * rnd.choice([0, 0, 1, 5])
        small = rnd.random() < 0.3
        rows = bytes(rnd.getrandbits(8) if not (small and (i % 32) < 28) else 0 for i in range(stride * n_rows))
        got = fb.commit_host(rows, n_rows, n_coeffs, row_stride=stride)
        want = common.oracle_commit_batch(curve, rows, n_rows, n_coeffs, bytes(base), row_stride=stride)
        cases += 1
        if got != want:
            fails += 1
            print("MISMATCH", curve, "base=%d c=%d rows=%d coeffs=%d stride=%d" % (nb, c, n_rows, n_coeffs, stride), flush=True)
    fb.close()
print("fuzz commit: %d cases, %d mismatches" % (cases, fails))
sys.exit(1 if fails else 0)
