#!/usr/bin/env python3
"""PCIe-inclusive latency of the reference's own boundary: compute_multi_exp(scalars, points, n) with HOST buffers (pageable
memory, as the unmodified Server passes them), next to the device-resident call.  Not bench.py's `value`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ctypes
from porla_amd import multiexp as mx, lib
from porla_amd.multiexp import _slice
from tests import common
if len(sys.argv) > 1 and sys.argv[1] == "--json":
    # one line for bench.py's `host_boundary` object: compute_multi_exp on caller-owned pageable host buffers at 2^log2n pairs
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n = 1 << log2n
    sc, pt = common.cached_inputs(n)
    bs, bp, out = ctypes.create_string_buffer(sc, 32 * n), ctypes.create_string_buffer(pt, 64 * n), ctypes.create_string_buffer(64)
    ss, sp, so = _slice(bs), _slice(bp), _slice(out)
    for _ in range(3):
        lib.compute_multi_exp(ctypes.byref(ss), ctypes.byref(sp), n, ctypes.byref(so))
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.compute_multi_exp(ctypes.byref(ss), ctypes.byref(sp), n, ctypes.byref(so))
    ms = (time.perf_counter() - t0) / reps * 1e3
    shards, devs = mx.last_msm_multi()
    print(json.dumps({"entry": "compute_multi_exp(host scalars, host points, n, out) -- porla/main.go:118-138", "ms": round(ms, 3),
                      "Mmul_s": round(n / ms / 1e3, 1), "pair_ranges": shards, "devices": devs, "result": out.raw.hex(),
                      "note": "96 n bytes cross PCIe inside the call; ranges are uploaded under the kernels of the previous range"}))
    sys.exit(0)
sc, pt = common.cached_inputs(1 << 20)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
for n in (3200, 1 << 14, 1 << 17, 1 << 20):
    # the caller's buffers exist before the call, as in the reference (no Python-side copies inside the timed region)
    bs, bp, out = ctypes.create_string_buffer(sc[:32 * n], 32 * n), ctypes.create_string_buffer(pt[:64 * n], 64 * n), ctypes.create_string_buffer(64)
    ss, sp, so = _slice(bs), _slice(bp), _slice(out)
    reps = 10
    for _ in range(3):
        lib.compute_multi_exp(ctypes.byref(ss), ctypes.byref(sp), n, ctypes.byref(so))
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.compute_multi_exp(ctypes.byref(ss), ctypes.byref(sp), n, ctypes.byref(so))
    t_host = (time.perf_counter() - t0) / reps * 1e3
    r_host = out.raw
    for _ in range(3):
        r_dev = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    t0 = time.perf_counter()
    for _ in range(reps):
        r_dev = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    t_dev = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps({"n": n, "host_buffers_ms": round(t_host, 3), "device_resident_ms": round(t_dev, 3), "same_result": r_host == r_dev,
                      "input_MiB": round(96 * n / 2**20, 2), "host_path_GBps": round(96 * n / (max(t_host - t_dev, 1e-6) * 1e-3) / 1e9, 1)}), flush=True)

# range-sharded host path (porla_bn254_msm_host_multi): ranges per device at 2^20 pairs, one device
n = 1 << 20
bs, bp = ctypes.create_string_buffer(sc[:32 * n], 32 * n), ctypes.create_string_buffer(pt[:64 * n], 64 * n)
out = ctypes.create_string_buffer(64)
want = None
for shards in (1, 2, 3, 4, 6, 8, 12, 16):
    for _ in range(3):
        lib.porla_bn254_msm_host_multi(bs, bp, n, shards, 1, out)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        lib.porla_bn254_msm_host_multi(bs, bp, n, shards, 1, out)
    el = (time.perf_counter() - t0) / reps * 1e3
    want = want or out.raw
    print(json.dumps({"n": n, "host_multi_shards": shards, "devices": 1, "ms": round(el, 3), "Mmul_s": round(n / el / 1e3, 1),
                      "same_result": out.raw == want}), flush=True)
