import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from porla_amd import multiexp as mx
from tests import common
n, distinct = 1 << 24, 1 << 14
pt = common.synth_points(distinct) * (n // distinct)
g = torch.Generator(device="cuda").manual_seed(24)
d_sc = torch.randint(0, 256, (32 * n,), dtype=torch.uint8, device="cuda", generator=g)
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
got = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
t0 = time.perf_counter(); got2 = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s); el = time.perf_counter() - t0
parts = b""
for k in range(8):
    lo = k * (n // 8)
    parts += mx.msm_device("bn254", d_sc.data_ptr() + 32 * lo, d_pt.data_ptr() + 64 * lo, n // 8, s, partial=True)
shape = mx.last_msm_shape()
print("2^24 pairs: %.2f ms, %.1f Mmul/s (window bits, windows, GLV of the last range MSM: %s); 8-way range fold equal: %s; repeat equal: %s" % (el * 1e3, n / el / 1e6, shape, mx.jac_sum("bn254", parts, 8) == got, got == got2))
t0 = time.perf_counter(); want = common.oracle_msm(bytes(d_sc.cpu().numpy()), pt, n); print("oracle %.1f s, equal: %s" % (time.perf_counter() - t0, want == got))
