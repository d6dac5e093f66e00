import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from porla_amd import multiexp as mx, lib
from tests import common
N = 1 << 18
sc, pt = common.cached_inputs(N)
d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
s = torch.cuda.current_stream().cuda_stream
for curve in ("bn254", "secp256k1"):
    for lg in (13, 14, 15, 16, 17, 18):
        n = 1 << lg
        row = {"curve": curve, "log2n": lg}
        ref = None
        for glv in (0, 1):
            lib.porla_gpu_set_msm_glv(glv)
            for _ in range(3):
                r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
            ref = ref or r
            t0 = time.perf_counter()
            for _ in range(20):
                r = mx.msm_device(curve, d_sc.data_ptr(), d_pt.data_ptr(), n, s)
            row["glv%d_ms" % glv] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
            row["shape%d" % glv] = mx.last_msm_shape()
            row["same"] = r == ref
        lib.porla_gpu_set_msm_glv(-1)
        print(json.dumps(row), flush=True)
