import os, sys, time, json
sys.path.insert(0, "/root/repo")
import torch
from porla_amd import icc, multiexp as mx, lib
from tests import common
for mode, mmax in (("matrix", 2048), ("ladder", 0)):
    lib.porla_icc_mac_set_matrix_max(mmax)
    for logn in (4, 6, 8, 9, 10, 11):
        n = 1 << logn
        base = common.synth_points(min(n, 4096), start=9000)
        macs = base[:64 * n]
        d_in = torch.frombuffer(bytearray(macs), dtype=torch.uint8).cuda()
        d_out = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            icc.mac_crebuild_device(d_in.data_ptr(), n, "bn254", 0, 0, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            icc.mac_crebuild_device(d_in.data_ptr(), n, "bn254", 0, 0, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        print(mode, "n=2^%d" % logn, round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms", flush=True)
lib.porla_icc_mac_set_matrix_max(0)
