#!/usr/bin/env python3
"""ms per round of the IPA prover's L / R points as ONE two-row fixed-base commitment over generators[0..127] || u (secp256k1;
tests/test_ipa_prover_rounds_gpu.py has the mapping to Server::inner_product_prove, Server.hpp:2318-2443), and the six rounds of a
proof one after the other (each round's rows depend on the previous round's points through the transcript hash)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from porla_amd import multiexp as mx
from tests import common
import random
N = 129
gens = common.secp_bench_points(N)
fb = mx.FixedBase("secp256k1", gens, N)
rnd = random.Random(1)
Q = common.SECP_N
def rows(n_terms):
    out = []
    for _ in range(2):
        r = [0] * N
        for j in rnd.sample(range(128), n_terms):
            r[j] = rnd.randrange(Q)
        r[128] = rnd.randrange(Q)
        out.append(b"".join(v.to_bytes(32, "big") for v in r))
    return b"".join(out)
cases = [rows(64) for _ in range(6)]
for c in cases:
    fb.commit_host(c, 2, N)
t0 = time.perf_counter()
reps = 200
for _ in range(reps):
    for c in cases:
        fb.commit_host(c, 2, N)
ms = (time.perf_counter() - t0) / reps * 1e3
print(json.dumps({"what": "IPA prover: L and R of a round as a 2-row commitment over 129 fixed secp256k1 points (64 + 1 non-zero coefficients per row)",
                  "ms_per_round": round(ms / 6, 4), "ms_per_six_rounds": round(ms, 4), "table": fb.info()}))
