#!/usr/bin/env python3
"""k_digits_partition: blocks per tile (PORLA_DIGITS_WG) against input size -- blocking latency and the kernel's own time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx
from tests import common
N = 1 << 20
sc, pt = common.cached_inputs(N)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
for lg in (13, 14, 16, 17, 18, 19, 20):
    n = 1 << lg
    ref = None
    for wg in ("", "1", "2", "4", "8", "16"):
        if wg: os.environ["PORLA_DIGITS_WG"] = wg
        else: os.environ.pop("PORLA_DIGITS_WG", None)
        for _ in range(3):
            r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        ref = ref or r
        mx.profile_enable(True)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        el = (time.perf_counter() - t0) / reps
        prof = {k: ms / max(c, 1) for k, ms, c in mx.profile_get()}
        mx.profile_enable(False)
        print(json.dumps({"log2n": lg, "wg": wg or "default", "ms": round(el * 1e3, 4), "digits_ms": round(prof.get("digits_partition", 0), 4),
                          "same": r == ref}), flush=True)
