#!/usr/bin/env python3
"""Latency of the reference's own call pattern: one compute_digest_from_srs / create_proof / verify_proof at a time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from porla_amd import multiexp as mx
from tests import common
tau = bytes.fromhex("ffeeddccbbaa99887766554433221100"); alpha = bytes.fromhex("00112233445566778899aabbccddeeff")
mx.init_key(tau, alpha)
blob = mx.init_SRS(128); mx.init_SRS_from_data(128, blob)
data = common.synth_scalars(128, start=5)
mx.compute_digest_from_srs(data)
for name, fn in (("compute_digest_from_srs", lambda: mx.compute_digest_from_srs(data)),
                 ("create_proof", lambda: mx.create_proof(12345, data)),
                 ("compute_digest (host)", lambda: mx.compute_digest(data)),
                 ("mult_point (host)", lambda: mx.bn254_mult(mx.compute_digest(data), tau.rjust(32, b"\0")))):
    fn()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    print(json.dumps({"call": name, "latency_ms": round((time.perf_counter() - t0) / 50 * 1e3, 4)}), flush=True)
c, h, z, y = mx.create_proof(12345, data)
t0 = time.perf_counter()
for _ in range(10):
    ok = mx.verify_proof(c, h, z, y)
print(json.dumps({"call": "verify_proof (host pairing)", "latency_ms": round((time.perf_counter() - t0) / 10 * 1e3, 3), "ok": ok}))
