#!/usr/bin/env python3
"""Audit row combine (porla_audit_combine_device; Server::audit's B += coeff * row loop, Server.hpp:790-828, + align_MAC's scalar
part on B, :903 -> :531-541) timed on a level store resident in HBM: wall ms per call, the two kernels' HIP-event times, and the
HBM roofline -- this is the one kernel of the path that HBM bounds (8 192 algorithmic bytes per challenged 64-byte-symbol row,
4 096 per 32-byte-symbol row).  The reference's sizes (NUM_CHECK_AUDIT * height <= 3 200 rows) and larger challenges for the
bandwidth regime.  A 2^10-row challenge is checked against the Python restatement (oracle/icc_py.py) in the same run.

    python tools/bench_audit.py [log2_store_rows=17] [challenge sizes, comma separated]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from porla_amd import icc, multiexp as mx

HBM_PEAK_GBPS = 8000.0
N_COLS = 128


def measure(log_store, sizes, mixed=True, reps_small=200):
    """-> (list of per-size dicts, cpu_baseline dict); the 1 024-row parity check rides in every dict"""
    store_rows = 1 << log_store
    g = torch.Generator(device="cuda").manual_seed(11)
    # the store: 64-byte little-endian symbols < LCM (510 bits for the KZG build: top byte < 0x20 keeps every value below it)
    d_s64 = torch.randint(0, 256, (store_rows, N_COLS, 64), dtype=torch.uint8, device="cuda", generator=g)
    d_s64[:, :, 63] &= 0x1f
    d_s32 = torch.randint(0, 256, (store_rows // 4, N_COLS, 32), dtype=torch.uint8, device="cuda", generator=g)
    d_s32[:, :, 31] &= 0x3f                                  # < 2^254 < p_icc = 207 * 2^248 + 1
    stream = torch.cuda.current_stream().cuda_stream
    outs = [torch.empty(sz * N_COLS, dtype=torch.uint8, device="cuda") for sz in (80, 32, 32, 32)]
    rng = np.random.Generator(np.random.PCG64(5))

    def challenge(n64, n32):
        i64 = torch.from_numpy(rng.integers(0, store_rows, max(n64, 1), dtype=np.int64)).cuda()
        i32 = torch.from_numpy(rng.integers(0, store_rows // 4, max(n32, 1), dtype=np.int64)).cuda()
        c64 = torch.from_numpy(rng.integers(0, 1 << 31, max(n64, 1), dtype=np.int64).astype(np.int32)).cuda()
        c32 = torch.from_numpy(rng.integers(0, 1 << 31, max(n32, 1), dtype=np.int64).astype(np.int32)).cuda()
        return i64, c64, i32, c32

    def call(ch, n64, n32):
        i64, c64, i32, c32 = ch
        icc.audit_combine_device(d_s64.data_ptr(), i64.data_ptr(), c64.data_ptr(), n64, d_s32.data_ptr(), i32.data_ptr(), c32.data_ptr(),
                                 n32, N_COLS, "bn254", outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), stream)

    # parity of one challenge against the Python restatement (exact sums, rows mod p_icc, alignment scalars)
    import icc_py
    n64, n32 = 768, 256
    ch = challenge(n64, n32)
    call(ch, n64, n32)
    torch.cuda.synchronize()
    rows = [[int.from_bytes(bytes(r[j].tolist()), "little") for j in range(N_COLS)] for r in d_s64[ch[0]].cpu().numpy()]
    rows += [[int.from_bytes(bytes(r[j].tolist()), "little") for j in range(N_COLS)] for r in d_s32[ch[2]].cpu().numpy()]
    coefs = [int(x) for x in ch[1].cpu().numpy().view(np.uint32)] + [int(x) for x in ch[3].cpu().numpy().view(np.uint32)]
    t0 = time.perf_counter()
    B, mods, cs = icc_py.audit_combine(rows, coefs, "bn254")
    cpu_s = time.perf_counter() - t0
    cpu = {"value": round((n64 + n32) / cpu_s, 1), "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": "the %d-row parity challenge through oracle/icc_py.py:audit_combine (Python integers, not NTL), %.2f s" % (n64 + n32, cpu_s)}
    ex, al, sc = (bytes(outs[k].cpu().numpy()) for k in (0, 1, 3))
    ok = all(int.from_bytes(ex[80 * j:80 * j + 80], "little") == B[j] and int.from_bytes(al[32 * j:32 * j + 32], "little") == mods[j]
             and int.from_bytes(sc[32 * j:32 * j + 32], "big") == cs[j] for j in range(N_COLS))

    res = []
    for n in sizes:
        for label, n64, n32 in ((("rows64", n, 0), ("rows64+rows32", n - n // 2, n // 2)) if mixed else (("rows64", n, 0),)):
            ch = challenge(n64, n32)
            for _ in range(5):
                call(ch, n64, n32)
            torch.cuda.synchronize()
            reps = reps_small if n <= 32768 else 20
            t0 = time.perf_counter()
            for _ in range(reps):
                call(ch, n64, n32)
            torch.cuda.synchronize()
            wall_ms = (time.perf_counter() - t0) / reps * 1e3
            mx.profile_enable(True)
            for _ in range(10):
                call(ch, n64, n32)
            torch.cuda.synchronize()
            prof = {k: ms / cnt for k, ms, cnt in mx.profile_get()}
            mx.profile_enable(False)
            alg = 8192 * n64 + 4096 * n32 + 64 * N_COLS + 12 * n
            k_ms = prof.get("audit_accumulate", 0.0)
            res.append({
                "workload": "audit row combine, %d challenged rows (%s) of a %d-row store x 128 symbols" % (n, label, store_rows),
                "rows": n, "ms_per_call_back_to_back": round(wall_ms, 4), "rows_per_s": round(n / wall_ms * 1e3, 1),
                "kernels_ms": {k: round(v, 4) for k, v in prof.items()},
                "roofline": {"bound": "hbm", "kernel": "k_audit_accumulate", "achieved": round(alg / k_ms / 1e6, 1) if k_ms else None,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / k_ms / 1e6 / HBM_PEAK_GBPS, 4) if k_ms else None,
                             "traffic": None, "algorithmic_bytes": alg},
                "bit_exact_vs_oracle_1024_row_challenge": ok})
    del d_s64, d_s32
    torch.cuda.empty_cache()
    return res, cpu


def main():
    log_store = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "128,1408,3200,32768,262144,1048576").split(",")]
    res, cpu = measure(log_store, sizes)
    for r in res:
        r["cpu_baseline"] = cpu
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
