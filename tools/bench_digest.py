#!/usr/bin/env python3
"""Client::initialize's block MACs in batches (porla_kzg_digest_batch_device = compute_digest hoisted over rows, main.go:70-89,
Client.hpp:408-419; porla_kzg_complement_batch_device, main.go:91-101, Client.hpp:445-455; porla_kzg_mac_batch_device = both and the
add_point that joins them, Client.hpp:229-236) on device-resident blocks: ms per batch,
blocks/s, per-kernel HIP-event times and the HBM roofline of the evaluation kernel (4 096 algorithmic bytes in + 32 out per block).
Rows 0, 1 and the last are checked against the library's own one-row symbol compute_digest (itself checked against the oracle by
tests/test_fixed_base_gpu.py).

    python tools/bench_digest.py [log2 row counts, comma separated]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx

HBM_PEAK_GBPS = 8000.0


def main():
    logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,14,17,19").split(",")]
    mx.init_key(bytes(range(1, 17)), bytes(range(17, 33)))
    mx.init_SRS(128)
    g = torch.Generator(device="cuda").manual_seed(7)
    s = torch.cuda.current_stream().cuda_stream
    for lg in logs:
        n = 1 << lg
        d_rows = torch.randint(0, 256, (n, 4096), dtype=torch.uint8, device="cuda", generator=g)
        d_out = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        d_sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        d_sc[:, :16] = 0                              # 16-byte PRF outputs left-padded to 32 bytes
        res = {"blocks": n}
        for name, fn in (("digest", lambda: mx.kzg_digest_batch_device(d_rows.data_ptr(), n, d_out.data_ptr(), s)),
                         ("complement", lambda: mx.kzg_complement_batch_device(d_sc.data_ptr(), n, d_out.data_ptr(), s)),
                         ("mac", lambda: mx.kzg_mac_batch_device(d_rows.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), s))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if name == "digest":
                got = bytes(d_out.cpu().numpy())
                ok = all(got[64 * r:64 * r + 64] == mx.compute_digest(bytes(d_rows[r].cpu().numpy())) for r in (0, 1, n - 1))
                res["bit_exact_vs_compute_digest_rows_0_1_last"] = ok
            if name == "mac":           # digest + complement joined on the host, three blocks
                got = bytes(d_out.cpu().numpy())
                res["mac_bit_exact_vs_digest_plus_complement_rows_0_1_last"] = all(
                    got[64 * r:64 * r + 64] == mx.bn254_add(mx.compute_digest(bytes(d_rows[r].cpu().numpy())),
                                                            mx.compute_digest_complement(bytes(d_sc[r, 16:].cpu().numpy())))
                    for r in (0, 1, n - 1))
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            mx.profile_enable(True)
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            prof = {k: round(t / c, 4) for k, t, c in mx.profile_get()}
            mx.profile_enable(False)
            res[name] = {"ms_per_batch": round(ms, 4), "M_blocks_per_s": round(n / ms / 1e3, 3), "kernels_ms": prof}
            if name == "digest" and prof.get("kzg_eval_rows"):
                alg = n * (4096 + 32)
                res[name]["roofline"] = {"bound": "hbm", "kernel": "k_kzg_eval_rows", "achieved": round(alg / prof["kzg_eval_rows"] / 1e6, 1),
                                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / prof["kzg_eval_rows"] / 1e6 / HBM_PEAK_GBPS, 4)}
        print(json.dumps(res), flush=True)
        del d_rows, d_out, d_sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
