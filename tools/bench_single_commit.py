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

# the reference's call pattern under load: T threads each calling compute_digest_from_srs one row at a time (the pool of 8,
# Server.hpp:1054-1078); calls that meet inside the library are coalesced into one launch
import threading
import ctypes
from porla_amd.loader import lib, GoSlice
def worker(count, seed, errs):
    row = ctypes.create_string_buffer(common.synth_scalars(128, start=seed), 4096)
    out = ctypes.create_string_buffer(64)
    si, so = GoSlice(ctypes.cast(row, ctypes.c_void_p), 4096, 4096), GoSlice(ctypes.cast(out, ctypes.c_void_p), 64, 64)
    want = None
    for _ in range(count):
        lib.compute_digest_from_srs(ctypes.byref(si), ctypes.byref(so))     # ctypes releases the GIL during the call
        if want is None:
            want = out.raw
        elif out.raw != want:
            errs.append(seed)
for T in (1, 2, 4, 8, 16):
    errs = []
    count = 400
    th = [threading.Thread(target=worker, args=(count, 1000 * t, errs)) for t in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print(json.dumps({"call": "compute_digest_from_srs", "threads": T, "commits_per_s": round(T * count / el, 1),
                      "latency_ms_per_call": round(el / count * 1e3, 4), "consistent": not errs}), flush=True)
