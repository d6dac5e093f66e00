#!/usr/bin/env python3
"""Sweep of the batched fixed-base KZG commitment (porla_fixed_base_*): commits/s by table window and batch size.
Rows are resident in HBM; times are HIP-event kernel times from the library plus the wall time of the whole call."""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", default="8,12,14,16")
    ap.add_argument("--rows", default="1,1024,131072,1048576")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from porla_amd import multiexp as mx
    from tests import common
    o = common.oracle()
    tau = bytes.fromhex("ffeeddccbbaa99887766554433221100")
    alpha = bytes.fromhex("00112233445566778899aabbccddeeff")
    o.oracle_kzg_init_key(tau, ctypes.c_size_t(16), alpha, ctypes.c_size_t(16))
    o.oracle_kzg_init_srs(ctypes.c_size_t(128), (1).to_bytes(32, "big"))
    raw = ctypes.create_string_buffer(64 * 128)
    o.oracle_kzg_srs_g1_raw(raw)
    srs = raw.raw
    max_rows = max(int(r) for r in args.rows.split(","))
    g = torch.Generator(device="cuda").manual_seed(1)
    d_rows = torch.randint(0, 256, (max_rows * 4096,), dtype=torch.uint8, device="cuda", generator=g)
    d_out = torch.empty(max_rows * 64, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for c in [int(x) for x in args.windows.split(",")]:
        t0 = time.perf_counter()
        mx.profile_enable(True)
        fb = mx.FixedBase("bn254", srs, 128, window_bits=c)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        build_prof = {k: round(ms, 3) for k, ms, _ in mx.profile_get()}
        info = fb.info()
        for n_rows in [int(r) for r in args.rows.split(",")]:
            fb.commit_device(d_rows.data_ptr(), n_rows, 128, d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            mx.profile_enable(True)
            t1 = time.perf_counter()
            for _ in range(args.reps):
                fb.commit_device(d_rows.data_ptr(), n_rows, 128, d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t1) / args.reps
            prof = {k: ms / max(cnt, 1) for k, ms, cnt in mx.profile_get()}
            mx.profile_enable(False)
            # check 2 rows against the oracle
            got = bytes(d_out[:128].cpu().numpy())
            rows2 = bytes(d_rows[:8192].cpu().numpy())
            ok = got == common.oracle_commit_batch("bn254", rows2, 2, 128, srs) if n_rows >= 2 else \
                got[:64] == common.oracle_commit_batch("bn254", rows2, 1, 128, srs)
            print(json.dumps({"window_bits": info["window_bits"], "windows": info["windows"],
                              "table_GB": round(info["table_bytes"] / 1e9, 3), "build_s": round(build_s, 3),
                              "build_kernels_ms": build_prof, "rows": n_rows, "wall_ms": round(wall * 1e3, 4),
                              "commits_per_s": round(n_rows / wall, 1),
                              "kernel_ms": {k: round(v, 4) for k, v in prof.items()}, "bit_exact_2rows": ok}), flush=True)
        fb.close()


if __name__ == "__main__":
    main()
