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

# ---- the IPA build's audit up to the proof, in one call (porla_ipa_audit_device): synthetic level store and MAC arrays
import numpy as np
import torch
NC, ROWS = 128, 1 << 12
g = torch.Generator(device="cuda").manual_seed(5)
d_s64 = torch.randint(0, 256, (ROWS, NC, 64), dtype=torch.uint8, device="cuda", generator=g)
d_s64[:, :, 63] &= 0x3f                                   # < 2^510 < LCM of the IPA build (512 bits)
pts = common.secp_bench_points(ROWS + 1)
d_ms = torch.frombuffer(bytearray(pts[:64 * ROWS]), dtype=torch.uint8).cuda()
d_as = torch.frombuffer(bytearray(pts[64:64 * (ROWS + 1)]), dtype=torch.uint8).cuda()
fb_g = mx.FixedBase("secp256k1", gens[:64 * NC], NC)
rng = np.random.Generator(np.random.PCG64(3))
for m in (1408, 3200):
    d_i = torch.from_numpy(rng.integers(0, ROWS, m, dtype=np.int64)).cuda()
    d_c = torch.from_numpy(rng.integers(0, 1 << 31, m, dtype=np.int64).astype(np.int32)).cuda()
    torch.cuda.synchronize()
    call = lambda: fb_g.ipa_audit_device(d_s64.data_ptr(), d_i.data_ptr(), d_c.data_ptr(), m, 0, 0, 0, 0, NC, d_ms.data_ptr(), d_as.data_ptr(),
                                         d_i.data_ptr(), d_c.data_ptr(), m)
    for _ in range(10):
        r = call()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        call()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    idx, coef = d_i.cpu().numpy(), d_c.cpu().numpy().view(np.uint32)
    sc = b"".join(int(c).to_bytes(32, "big") for c in coef)
    ok = r["combined_mac"] == common.oracle_secp_msm(sc, b"".join(pts[64 * int(i):64 * int(i) + 64] for i in idx), m)
    print(json.dumps({"what": "IPA audit up to the proof in one call (porla_ipa_audit_device), %d challenged rows" % m,
                      "ms_per_audit_median": round(ts[100], 4), "p99_ms": round(ts[198], 4), "combined_mac_equals_oracle": ok}))
