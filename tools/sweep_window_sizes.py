#!/usr/bin/env python3
"""Wall time of one blocking MSM for every (size, window width) around the automatic choice: the data behind the window
model of msm_impl.hip.h:choose_window.  Usage: sweep_window_sizes.py [bn254|secp256k1] [full|int32]"""
import json, os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx, lib
from tests import common
curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
kind = sys.argv[2] if len(sys.argv) > 2 else "full"
N = 1 << 20
if curve == "bn254":
    sc, pt = common.cached_inputs(N)
else:
    sc, pt = common.secp_bench_scalars(N), common.secp_bench_points(N)
if kind == "int32":   # the audit's coefficients: abs(int32) as 32-byte big-endian scalars
    sc = b"".join(b"\0" * 28 + struct.pack(">I", int.from_bytes(sc[32 * i + 28:32 * i + 32], "big") & 0x7fffffff) for i in range(1 << 16))
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
sizes = (128, 1408, 3200, 1 << 14, 1 << 16) if kind == "int32" else (128, 1408, 3200, 1 << 14, 1 << 17, 1 << 20)
if len(sys.argv) > 3:   # explicit log2 sizes: sweep_window_sizes.py secp256k1 full 13,14,15,16,17,18,19
    sizes = tuple(1 << int(x) for x in sys.argv[3].split(","))

def run(n, reps):
    for _ in range(2):
        r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    return (time.perf_counter() - t0) / reps * 1e3, r

for n in sizes:
    lib.porla_gpu_set_msm_window(0)
    t_auto, ref = run(n, 10)
    c_auto, w_auto, glv = mx.last_msm_shape()
    row = {}
    for c in range(max(2, c_auto - 6), min(20, c_auto + 4) + 1):
        lib.porla_gpu_set_msm_window(c)
        t, r = run(n, 10)
        assert r == ref
        row[c] = round(t, 4)
    lib.porla_gpu_set_msm_window(0)
    best = min(row, key=row.get)
    print(json.dumps({"curve": curve, "scalars": kind, "n": n, "auto_c": c_auto, "auto_ms": round(t_auto, 4), "best_c": best,
                      "best_ms": row[best], "by_c": row}), flush=True)
