#!/usr/bin/env python3
"""A loop of blocking 2^log2n-pair BN254 MSMs on device-resident inputs (the caller that waits for every result): run it under
`rocprofv3 --kernel-trace --stats -- python3 tools/blocking_loop.py [log2n] [reps]` for per-kernel durations without the overlap of
bench.py's two MSMs in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx
from tests import common
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << log2n
sc, pt = common.cached_inputs(n)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
el = (time.perf_counter() - t0) / reps
print("blocking 2^%d MSM: %.4f ms, %.1f Mmul/s, shape %s" % (log2n, el * 1e3, n / el / 1e6, mx.last_msm_shape()))
if os.environ.get("PORLA_LOOP_PROFILE"):
    mx.profile_enable(True)
    for _ in range(5):
        mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    print("kernels (ms per MSM):", {k: round(ms / max(c, 1), 4) for k, ms, c in mx.profile_get()})
    mx.profile_enable(False)
