#!/usr/bin/env python3
"""Blocking 2^20 BN254 MSM: plain windows vs endomorphism split, window width swept, with the per-kernel breakdown."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx, lib
from tests import common
n = 1 << 20
sc, pt = common.cached_inputs(n)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
want = None
for glv in (0, 1):
    lib.porla_gpu_set_msm_glv(glv)
    for c in ((15, 16, 17, 18) if glv == 0 else (14, 15, 16, 17, 18, 19)):
        lib.porla_gpu_set_msm_window(c)
        for _ in range(3):
            r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        want = want or r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        el = (time.perf_counter() - t0) / reps
        mx.profile_enable(True)
        for _ in range(3):
            mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        prof = {k: round(ms / 3, 4) for k, ms, cnt in mx.profile_get()}
        mx.profile_enable(False)
        print(json.dumps({"glv": glv, "c": c, "shape": mx.last_msm_shape(), "blocking_ms": round(el * 1e3, 4), "same": r == want, "kernels_ms": prof}), flush=True)
lib.porla_gpu_set_msm_window(0)
lib.porla_gpu_set_msm_glv(-1)
