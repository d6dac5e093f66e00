#!/usr/bin/env python3
"""MAC-side encode timing (porla_icc_mac_encode_device): per-stage kernel time and MAC butterflies/s by N."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import icc, multiexp as mx
from tests import common
for logn in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,15,18").split(",")]:
    n = 1 << logn
    base = common.synth_points(min(n, 4096), start=9000)
    macs = (base * (n // min(n, 4096)))[:64 * n]
    d_in = torch.frombuffer(bytearray(macs), dtype=torch.uint8).cuda()
    d_out = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    icc.mac_crebuild_device(d_in.data_ptr(), n, "bn254", 0, 0, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    mx.profile_enable(True)
    t0 = time.perf_counter()
    icc.mac_crebuild_device(d_in.data_ptr(), n, "bn254", 0, 0, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    prof = {k: (round(ms, 3), cnt) for k, ms, cnt in mx.profile_get()}
    mx.profile_enable(False)
    print(json.dumps({"log2_n": logn, "wall_ms": round(wall * 1e3, 3), "scalar_mults": (n // 2) * logn,
                      "M_scalar_mult_per_s": round((n // 2) * logn / wall / 1e6, 3), "kernels_ms_total_and_launches": prof}), flush=True)
