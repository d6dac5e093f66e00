#!/usr/bin/env python3
"""Single-launch commitment path (k_fb_commit_small) by rows per batch: wall time of the host call and the kernel's own time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from porla_amd import multiexp as mx
TAU = bytes.fromhex("ffeeddccbbaa99887766554433221100"); ALPHA = bytes.fromhex("00112233445566778899aabbccddeeff")
mx.init_key(TAU, ALPHA)
mx.init_SRS_from_data(128, mx.init_SRS(128))
for rows in (1, 8, 32, 64, 65, 128, 256, 512, 1024, 1025, 2048, 4096, 8192):
    data = os.urandom(4096 * rows)
    for _ in range(3):
        r = mx.kzg_commit_batch_host(data, rows)
    t0 = time.perf_counter()
    for _ in range(50):
        r = mx.kzg_commit_batch_host(data, rows)
    wall = (time.perf_counter() - t0) / 50 * 1e3
    mx.profile_enable(True)
    mx.kzg_commit_batch_host(data, rows)
    prof = {k: round(ms / max(c, 1), 4) for k, ms, c in mx.profile_get()}
    mx.profile_enable(False)
    print(json.dumps({"rows": rows, "host_call_ms": round(wall, 4), "kernels_ms": prof}), flush=True)
