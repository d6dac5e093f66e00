#!/usr/bin/env python3
"""Latency of small MSMs (the reference's real audit sizes: 128 .. 3 200 pairs) through the blocking device-pointer call."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx
from tests import common
N = 1 << 14
sc, pt = common.synth_inputs(N)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
import random
rnd = random.Random(5)
audit_sc = b"".join(mx.bn254_scalar_set_int(rnd.getrandbits(31)) for _ in range(N))   # abs(int32), utils.h:271-275
d_audit = torch.frombuffer(bytearray(audit_sc), dtype=torch.uint8).cuda()
for n in (128, 1408, 3200, 1 << 14):
    for _ in range(3):
        r = mx.msm_device("bn254", d_audit.data_ptr(), d_pt.data_ptr(), n, s)
    t0 = time.perf_counter()
    for _ in range(20):
        r = mx.msm_device("bn254", d_audit.data_ptr(), d_pt.data_ptr(), n, s)
    el = (time.perf_counter() - t0) / 20
    print(json.dumps({"n": n, "scalars": "abs(int32)", "latency_ms": round(el * 1e3, 4), "ok": r == common.oracle_msm(audit_sc, pt, n)}), flush=True)
for n in (128, 1408, 3200, 1 << 14):
    for _ in range(3):
        r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    mx.profile_enable(True)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    el = (time.perf_counter() - t0) / reps
    prof = {k: round(ms / max(c, 1), 4) for k, ms, c in mx.profile_get()}
    mx.profile_enable(False)
    print(json.dumps({"n": n, "latency_ms": round(el * 1e3, 4), "ok": r == common.oracle_msm(sc, pt, n), "kernels_ms": prof}), flush=True)
