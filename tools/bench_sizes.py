#!/usr/bin/env python3
"""MSM wall time by size with the automatic window (both curves): checks the window model of msm_impl.hip.h:choose_window."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx
from tests import common
N = 1 << 20
s = torch.cuda.current_stream().cuda_stream
for curve in ("bn254", "secp256k1"):
    if curve == "bn254":
        sc, pt = common.cached_inputs(N)
    else:
        sc, pt = common.secp_bench_scalars(N), common.secp_bench_points(N)
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    for n in (128, 1408, 3200, 1 << 14, 1 << 17, 1 << 20):
        for _ in range(3):
            r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        el = (time.perf_counter() - t0) / reps
        print(json.dumps({"curve": curve, "n": n, "ms": round(el * 1e3, 4), "Mmul_s": round(n / el / 1e6, 2)}), flush=True)
