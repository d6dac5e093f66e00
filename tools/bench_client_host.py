#!/usr/bin/env python3
"""The client's block MACs from HOST buffers (porla_kzg_mac_batch_host: digest + complement + add_point per block, Client.hpp:216-236,
408-455), PCIe included: ms per batch and blocks/s by batch size, beside the device-resident figure of the same batch.  Three blocks
of every size are checked against the one-row host symbols.

    python tools/bench_client_host.py [log2 block counts, comma separated]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from porla_amd import multiexp as mx


def main():
    logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,14,17").split(",")]
    mx.init_key(bytes(range(1, 17)), bytes(range(17, 33)))
    mx.init_SRS(128)
    rng = np.random.default_rng(5)
    s = torch.cuda.current_stream().cuda_stream
    for lg in logs:
        n = 1 << lg
        rows = rng.integers(0, 256, size=n * 4096, dtype=np.uint8).tobytes()
        sc = np.zeros((n, 32), dtype=np.uint8)
        sc[:, 16:] = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
        sc = sc.tobytes()
        got = mx.kzg_mac_batch_host(rows, sc, n)
        ok = all(got[64 * r:64 * r + 64] == mx.bn254_add(mx.compute_digest(rows[4096 * r:4096 * r + 4096]),
                                                         mx.compute_digest_complement(sc[32 * r + 16:32 * r + 32])) for r in (0, 1, n - 1))
        reps = 20 if lg <= 14 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            mx.kzg_mac_batch_host(rows, sc, n)
        host_ms = (time.perf_counter() - t0) / reps * 1e3
        d_rows = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
        d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
        d_out = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t0) / 20 * 1e3
        print(json.dumps({"blocks": n, "host_buffers_ms_per_batch": round(host_ms, 3), "host_buffers_M_blocks_per_s": round(n / host_ms / 1e3, 3),
                          "host_buffers_GBps_in": round(n * 4128 / host_ms / 1e6, 2), "device_resident_ms_per_batch": round(dev_ms, 4),
                          "bit_exact_vs_one_row_symbols_rows_0_1_last": ok}), flush=True)


if __name__ == "__main__":
    main()
