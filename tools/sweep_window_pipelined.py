#!/usr/bin/env python3
"""Window width c at 2^log2n BN254 pairs, inputs resident: blocking calls (ms per MSM) and two MSMs in flight on two streams
(ms per step, as bench.py's headline), without HIP events, then the per-kernel breakdown of blocking calls.
    python tools/sweep_window_pipelined.py [log2n] [c,c,...] [reps] [glv: -1 per-curve default | 0 | 1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx, lib
from tests import common
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [15, 16, 17, 18]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
glv = int(sys.argv[4]) if len(sys.argv) > 4 else -1
lib.porla_gpu_set_msm_glv(glv)            # BN254's default is no endomorphism split; 1 forces it (half the windows, twice the entries)
n = 1 << log2n
sc, pt = common.cached_inputs(n)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
streams = [torch.cuda.Stream() for _ in range(2)]
ref = None
for c in cs:
    lib.porla_gpu_set_msm_window(c)
    for _ in range(3):
        r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    ref = ref or r
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    blocking = (time.perf_counter() - t0) / reps * 1e3
    inflight = []
    k = 0
    def step():
        global k
        res = None
        if len(inflight) == 2:
            res = mx.msm_end(inflight.pop(0))
        slot = 1 + k % 2
        mx.msm_begin(slot, d_sc.data_ptr(), d_pt.data_ptr(), n, streams[k % 2].cuda_stream)
        inflight.append(slot)
        k += 1
        return res
    for _ in range(6):
        step()
    while inflight:
        mx.msm_end(inflight.pop(0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    while inflight:
        r2 = mx.msm_end(inflight.pop(0))
    torch.cuda.synchronize()
    piped = (time.perf_counter() - t0) / reps * 1e3
    mx.profile_enable(True)
    for _ in range(5):
        mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    torch.cuda.synchronize()
    prof = {kk: round(ms / 5, 4) for kk, ms, cnt in mx.profile_get()}
    mx.profile_enable(False)
    print(json.dumps({"c": c, "glv": glv, "shape": mx.last_msm_shape(), "blocking_ms": round(blocking, 4), "pipelined_ms_per_step": round(piped, 4),
                      "pipelined_Mmul_s": round(n / piped / 1e3, 1), "same_result": r == ref and r2 == ref,
                      "kernels_ms_per_msm": prof}), flush=True)
lib.porla_gpu_set_msm_window(0)
lib.porla_gpu_set_msm_glv(-1)
