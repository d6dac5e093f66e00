#!/usr/bin/env python3
"""porla_kzg_commit_batch_host on pageable host rows (the reference-side call of INTEGRATION.md s3: num_blocks x
compute_digest_from_srs in one call), PCIe included: ms per call and commits/s by batch size; first and last row checked against
the one-row symbol compute_digest_from_srs.

    python tools/bench_commit_host.py [log2 row counts, comma separated]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from porla_amd import multiexp as mx


def main():
    logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,14,17").split(",")]
    mx.init_key(bytes(range(1, 17)), bytes(range(17, 33)))
    blob = mx.init_SRS(128)
    mx.init_SRS_from_data(128, blob)
    rng = np.random.default_rng(9)
    for lg in logs:
        n = 1 << lg
        rows = rng.integers(0, 256, size=n * 4096, dtype=np.uint8).tobytes()
        got = mx.kzg_commit_batch_host(rows, n)
        ok = all(got[64 * r:64 * r + 64] == mx.compute_digest_from_srs(rows[4096 * r:4096 * r + 4096]) for r in (0, n - 1))
        reps = 10 if lg <= 14 else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            mx.kzg_commit_batch_host(rows, n)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"rows": n, "ms_per_call": round(ms, 3), "M_commits_per_s": round(n / ms / 1e3, 3),
                          "GBps_in": round(n * 4096 / ms / 1e6, 2), "bit_exact_vs_compute_digest_from_srs_rows_0_last": ok}), flush=True)


if __name__ == "__main__":
    main()
