#!/usr/bin/env python3
"""Runs a few blocking MSMs of one size (to be wrapped in rocprofv3 --kernel-trace).  Usage: trace_msm.py n [bn254|secp256k1] [full|int32]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx
from tests import common
n = int(sys.argv[1])
curve = sys.argv[2] if len(sys.argv) > 2 else "bn254"
kind = sys.argv[3] if len(sys.argv) > 3 else "full"
N = max(n, 1 << 12)
if curve == "bn254":
    sc, pt = common.cached_inputs(1 << 20)
else:
    sc, pt = common.secp_bench_scalars(N), common.secp_bench_points(N)
sc, pt = sc[:32 * N], pt[:64 * N]
if kind == "int32":
    sc = b"".join(b"\0" * 28 + struct.pack(">I", int.from_bytes(sc[32 * i + 28:32 * i + 32], "big") & 0x7fffffff) for i in range(N))
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
print(mx.last_msm_shape())
