import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from porla_amd import multiexp as mx
from tests import common
sc, pt = common.cached_inputs(1 << 15)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda(); d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
for n in (2048, 4096, 4097, 8192, 16384, 16385, 20000, 24576, 32768):
    for _ in range(3): r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    t0 = time.perf_counter()
    for _ in range(20): r = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
    print(json.dumps({"n": n, "ms": round((time.perf_counter() - t0) / 20 * 1e3, 4), "shape": mx.last_msm_shape()}), flush=True)
