#!/usr/bin/env python3
"""MSM window-width sweep at 2^20 pairs (inputs resident): per-kernel HIP-event times for each c."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx, lib
from tests import common
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [13, 14, 15, 16, 17, 18]
curve = sys.argv[3] if len(sys.argv) > 3 else "bn254"
lib.porla_gpu_set_msm_glv(int(sys.argv[4]) if len(sys.argv) > 4 else -1)
if curve == "bn254":
    sc, pt = common.cached_inputs(n)
else:
    sc, pt = common.secp_bench_scalars(n), common.secp_bench_points(n)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
ref = None
for c in cs:
    lib.porla_gpu_set_msm_window(c)
    for _ in range(2):
        r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    ref = ref or r
    torch.cuda.synchronize()
    mx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5
    prof = {k: round(ms / max(cnt, 1), 4) for k, ms, cnt in mx.profile_get()}
    mx.profile_enable(False)
    print(json.dumps({"c": c, "wall_ms": round(wall * 1e3, 4), "Mmul_s": round(n / wall / 1e6, 1), "same_result": r == ref,
                      "kernels_ms": prof, "sum_kernels": round(sum(prof.values()), 4)}), flush=True)
lib.porla_gpu_set_msm_window(0)
