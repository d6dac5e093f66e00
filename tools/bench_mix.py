#!/usr/bin/env python3
"""Server::mix data part on device-resident blocks (porla_icc_mix_device; Server.hpp:1269-1278): out[i] = (A0[i] + v^i A1[i]) % LCM,
out[i + len] = (A0[i] - v^i A1[i]) % LCM on rows of 128 symbols of 64 bytes.  Kernel time by block length, symbols/s, and the HBM
roofline (algorithmic bytes: 2 x 64 B in + 2 x 64 B out per butterfly); a 2^8-row mix is checked against oracle/icc_ref.c.

    python tools/bench_mix.py [log2 lengths, comma separated]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import lib, multiexp as mx
from tests import common

HBM_PEAK_GBPS = 8000.0
N_COLS = 128


def check(msg, rc):
    if rc:
        raise RuntimeError("%s: rc=%d" % (msg, rc))


def main():
    logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,10,12,14,16").split(",")]
    n_total = 1 << 17
    g = torch.Generator(device="cuda").manual_seed(3)
    vp = ctypes.c_void_p
    stream = torch.cuda.current_stream().cuda_stream
    ok = None
    for lg in logs:
        ln = 1 << lg
        blocks = []
        for _ in range(2):
            t = torch.randint(0, 256, (ln, N_COLS, 64), dtype=torch.uint8, device="cuda", generator=g)
            t[:, :, 63] &= 0x1f                      # < 2^509 < LCM (KZG build: 510 bits)
            blocks.append(t)
        d_out = torch.empty((2 * ln, N_COLS, 64), dtype=torch.uint8, device="cuda")

        def call():
            check("mix", lib.porla_icc_mix_device(vp(blocks[0].data_ptr()), vp(blocks[1].data_ptr()), ln, N_COLS, n_total, 0,
                                                  vp(d_out.data_ptr()), vp(stream)))
        call()
        torch.cuda.synchronize()
        if ok is None:          # the first (smallest) length against the C restatement
            a0, a1 = bytes(blocks[0].cpu().numpy()), bytes(blocks[1].cpu().numpy())
            want = ctypes.create_string_buffer(2 * ln * N_COLS * 64)
            common.oracle().oracle_icc_mix(a0, a1, ctypes.c_size_t(ln), ctypes.c_size_t(N_COLS), ctypes.c_size_t(n_total), 0, want)
            ok = bytes(d_out.cpu().numpy()) == want.raw
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        reps = 50 if lg <= 12 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) / reps * 1e3
        mx.profile_enable(True)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        prof = {k: ms / cnt for k, ms, cnt in mx.profile_get()}
        mx.profile_enable(False)
        k_ms = max(prof.values()) if prof else 0.0
        alg = 256 * ln * N_COLS
        print(json.dumps({"workload": "Server::mix data part, two blocks of 2^%d rows x 128 symbols -> 2^%d rows" % (lg, lg + 1),
                          "ms_per_call_back_to_back": round(wall_ms, 4), "G_symbols_out_per_s": round(2 * ln * N_COLS / wall_ms / 1e6, 3),
                          "kernels_ms": {k: round(v, 4) for k, v in prof.items()},
                          "roofline": {"bound": "hbm", "achieved": round(alg / k_ms / 1e6, 1) if k_ms else None, "peak": HBM_PEAK_GBPS,
                                       "unit": "GB/s", "frac": round(alg / k_ms / 1e6 / HBM_PEAK_GBPS, 4) if k_ms else None,
                                       "algorithmic_bytes": alg},
                          "bit_exact_vs_oracle_first_length": ok}), flush=True)
        del blocks, d_out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
