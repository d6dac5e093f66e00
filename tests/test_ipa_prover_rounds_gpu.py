"""GPU box: the L / R points of the IPA build's inner-product prover (Server::inner_product_prove, porla/Server/Server.hpp:2318-2443)
through the engine's fixed-base boundary.  Every round of the reference computes
    L = sum_{j in odd blocks}  (a[q] * x_values[j]) * generators[j] + cL * u
    R = sum_{j in even blocks} (a[half + q] * x_values[j]) * generators[j] + cR * u
as 8 pool threads x an 8-point secp256k1_ecmult_multi_var plus one secp256k1_ecmult_const.  The generators (and u) never change, so
both points of a round are TWO ROWS of one batched fixed-base commitment over the 129 points generators[0..127] || u
(porla_fixed_base_create once, porla_fixed_base_commit_host per round: coefficient j of row L is the scalar of generator j, zero on
the blocks L does not touch, coefficient 128 is cL) -- one launch per round instead of 16 MSM calls and 2 scalar multiplications.
The scalar bookkeeping (x_values, the folding of a and b, the Fiat-Shamir hash) stays the server's; it is restated here only to
drive the six rounds with the reference's index pattern.  Expected values: the oracle's secp256k1 MSM over the same scalars."""
import hashlib
import random

import pytest

from tests import common

pytestmark = pytest.mark.gpu
NUM_CHUNKS = 128
Q = common.SECP_N


def test_six_prover_rounds_as_two_row_commitments():
    from porla_amd import multiexp as mx
    gens = common.secp_bench_points(NUM_CHUNKS + 1)          # generators[0..127] and u
    fb = mx.FixedBase("secp256k1", gens, NUM_CHUNKS + 1)
    rnd = random.Random(2318)
    a = [rnd.randrange(Q) for _ in range(NUM_CHUNKS)]
    b = [rnd.randrange(Q) for _ in range(NUM_CHUNKS)]
    x_values = [1] * NUM_CHUNKS
    seed = hashlib.sha256(b"ipa").digest()
    half, k, rounds = NUM_CHUNKS // 2, 1, 0
    be = lambda v: v.to_bytes(32, "big")
    while half > 1:
        x = int.from_bytes(seed, "big") % Q or 1
        inv_x = pow(x, Q - 2, Q)
        cL = sum(a[i] * b[half + i] for i in range(half)) % Q
        cR = sum(a[half + i] * b[i] for i in range(half)) % Q
        row_l, row_r = [0] * (NUM_CHUNKS + 1), [0] * (NUM_CHUNKS + 1)
        for i in range(k):                                   # L: blocks 2i + 1 (Server.hpp:2341-2353)
            pos = 2 * i + 1
            for q, j in enumerate(range(pos * half, (pos + 1) * half)):
                row_l[j] = a[q] * x_values[j] % Q
                x_values[j] = x_values[j] * x % Q
        for i in range(k):                                   # R: blocks 2i (Server.hpp:2390-2402)
            pos = 2 * i
            for q, j in enumerate(range(pos * half, (pos + 1) * half)):
                row_r[j] = a[half + q] * x_values[j] % Q
                x_values[j] = x_values[j] * inv_x % Q
        row_l[NUM_CHUNKS], row_r[NUM_CHUNKS] = cL, cR
        rows = b"".join(be(v) for v in row_l) + b"".join(be(v) for v in row_r)
        got = fb.commit_host(rows, 2, NUM_CHUNKS + 1)
        for r, row in enumerate((row_l, row_r)):
            nz = [(v, j) for j, v in enumerate(row) if v]
            assert len(nz) <= NUM_CHUNKS // 2 + 1            # the reference's 64 generator terms + the u term
            sc = b"".join(be(v) for v, _ in nz)
            pt = b"".join(gens[64 * j:64 * j + 64] for _, j in nz)
            assert got[64 * r:64 * r + 64] == common.oracle_secp_msm(sc, pt, len(nz)), (rounds, r)
        seed = hashlib.sha256(seed + got).digest()           # (stands in for the reference's transcript hash)
        a = [(a[i] * x + a[i + half] * inv_x) % Q for i in range(half)]
        b = [(b[i] * inv_x + b[i + half] * x) % Q for i in range(half)]
        half >>= 1
        k <<= 1
        rounds += 1
    assert rounds == 6
