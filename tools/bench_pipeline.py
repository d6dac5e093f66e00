#!/usr/bin/env python3
"""Throughput of independent 2^20-pair BN254 MSMs with 1..3 in flight (two-phase API, one stream per slot)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx
from tests import common
n = 1 << 20
sc, pt = common.cached_inputs(n)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
want = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
steps = 24
for depth in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(depth)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        inflight = []
        ok = True
        for k in range(steps):
            slot = 1 + k % depth
            if len(inflight) == depth:
                ok &= mx.msm_end(inflight.pop(0)) == want
            mx.msm_begin(slot, d_sc.data_ptr(), d_pt.data_ptr(), n, streams[k % depth].cuda_stream)
            inflight.append(slot)
        while inflight:
            ok &= mx.msm_end(inflight.pop(0)) == want
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(json.dumps({"in_flight": depth, "ms_per_msm": round(el / steps * 1e3, 4), "Mmul_s": round(n * steps / el / 1e6, 1), "all_correct": ok}), flush=True)
