#!/usr/bin/env python3
"""bench_crebuild.py -- the last encode stage of a large CRebuild as the reference executes it, device-resident.

Reference pipeline (KZG build), porla/Server/Server.hpp:
  :1487-1536  copy the blocks of U, scale the Y part by wt, scale the Y MACs by wt
  :1548-1687  X part: 15 butterfly stages over 2^15 rows x 128 chunks (data) and over the 2^15 MACs ("FFT in the exponent")
  :1691-1830  Y part: the same on the scaled copies
  :1658-1676 / :2059-2065  last stage of each part: align_MAC(row) = the row mod p_icc (written as 256-bit values), the alignment
              scalars c = (A mod p_icc - A) mod q (:531-541) and MAC_alignments[row] += compute_digest_from_srs(c) (:550-560)
One STEP here = both parts for all rows, every arithmetic step on the engine, ONE stream, no host synchronisation inside:
  porla_icc_encode_device(part)       rows -> aligned rows (32 B / chunk) + alignment scalars (32 B big-endian / chunk)
  porla_kzg_commit_batch_device       alignment scalars -> one 64-byte commitment per row            (2 per row and step)
  porla_icc_mac_encode_device(part)   2^15 MACs -> encoded MACs
Inputs (raw 32-byte chunks, per-block MACs) and outputs stay in HBM.  Prints ONE JSON line in bench.py's format: value = rows
(blocks) per second through both parts; `kernels_ms` is the per-kernel split of a step, `hbm_bytes_per_row` the algorithmic
traffic; a 2^10-row sample of the same chain is checked against the oracle chain (oracle/icc_ref.c -> oracle/bn254_ref.c commit
batch, oracle/mac_ref.c), which also provides `cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TAU = bytes.fromhex("ffeeddccbbaa99887766554433221100")     # TAU_KEY, config.hpp:39
ALPHA = bytes.fromhex("00112233445566778899aabbccddeeff")   # SECRET_KEY, config.hpp:38
HBM_PEAK_GBPS = 8000.0
NCOLS = 128


def run_chain(torch, icc, mx, d_rows, d_macs, n, write_step, bufs, stream, mac_stream=None, reference_order=False):
    """both parts of the last encode stage for n rows; everything asynchronous, nothing waits on the host.
    reference_order: the calls as the reference orders its work (per part: encode, align_MAC commitments, MAC encode), one stream,
    one encode and one MAC encode per part.  Default: the data side AND the MAC halves of both parts each from ONE run of their
    network (porla_icc_encode_xy_device, porla_icc_mac_encode_xy_device: Y_k = wt * X_k), the MAC encode on `mac_stream` when given -- their stages are chains of dependent group operations on one wave per SIMD,
    the data side (encode + commitments) is bound by VALU issue: side by side they share the chip instead of queueing."""
    if reference_order:
        for part in (0, 1):
            al, sc, am, mh = bufs[part]
            icc.crebuild_device(d_rows.data_ptr(), n, NCOLS, "bn254", write_step, part, 0, al.data_ptr(), sc.data_ptr(), stream=stream)
            mx.kzg_commit_batch_device(sc.data_ptr(), n, am.data_ptr(), stream)
            icc.mac_crebuild_device(d_macs.data_ptr(), n, "bn254", write_step, part, mh.data_ptr(), stream)
    else:
        # the data side of both parts from one run of the network too (porla_icc_encode_xy_device: Y_k = wt X_k mod LCM)
        icc.crebuild_xy_device(d_rows.data_ptr(), n, NCOLS, "bn254", write_step, 0, bufs[0][0].data_ptr(), bufs[0][1].data_ptr(),
                               0, bufs[1][0].data_ptr(), bufs[1][1].data_ptr(), stream=stream)
        # both parts' alignment scalars as ONE batch of 2 n rows (they are contiguous: alloc()): one launch of the commitment kernel
        # instead of two -- the kernel that starts on the other stream right after such a launch was seen to take ~4 ms longer
        mx.kzg_commit_batch_device(bufs[0][1].data_ptr(), 2 * n, bufs[0][2].data_ptr(), stream)
        icc.mac_crebuild_xy_device(d_macs.data_ptr(), n, "bn254", write_step, bufs[0][3].data_ptr(), bufs[1][3].data_ptr(),
                                   mac_stream if mac_stream is not None else stream)


def alloc(torch, n, dev):
    """per part: (aligned rows, alignment scalars, their commitments, encoded MACs); the two parts' scalars lie back to back in one
    allocation, and so do their commitments: the 2 n rows are committed as ONE batch"""
    sc2 = torch.empty(2 * 32 * n * NCOLS, dtype=torch.uint8, device=dev)
    am2 = torch.empty(2 * 64 * n, dtype=torch.uint8, device=dev)
    return [(torch.empty(32 * n * NCOLS, dtype=torch.uint8, device=dev), sc2[part * 32 * n * NCOLS:(part + 1) * 32 * n * NCOLS],
             am2[part * 64 * n:(part + 1) * 64 * n], torch.empty(64 * n, dtype=torch.uint8, device=dev)) for part in (0, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2rows", type=int, default=int(os.environ.get("PORLA_CREBUILD_LOG2ROWS", "15")))
    ap.add_argument("--write-step", type=int, default=37)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    from porla_amd import icc, multiexp as mx
    from tests import common   # the oracle: input generation for the sample check and the cpu_baseline leg only

    assert torch.cuda.is_available(), "bench_crebuild.py needs a GPU (the engine has no CPU path)"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    n = 1 << args.log2rows
    mx.init_key(TAU, ALPHA)
    blob = mx.init_SRS(NCOLS)
    mx.init_SRS_from_data(NCOLS, blob)

    g = torch.Generator(device=dev).manual_seed(2024)
    d_rows = torch.randint(0, 256, (n * NCOLS * 32,), dtype=torch.uint8, device=dev, generator=g)
    # the blocks' MACs: commitments of the rows read as big-endian coefficients -- any 2^15 valid G1 points serve the benchmark
    d_macs = torch.empty(64 * n, dtype=torch.uint8, device=dev)
    mx.kzg_commit_batch_device(d_rows.data_ptr(), n, d_macs.data_ptr(), stream)
    bufs = alloc(torch, n, dev)
    torch.cuda.synchronize()

    # high priority: a MAC stage is one wave per SIMD walking a chain of dependent additions -- its blocks must get a CU slot as
    # soon as one frees up, the wide commitment grid takes what is left
    side = torch.cuda.Stream(device=dev, priority=-1)

    def step(mode):
        if mode == "stage_call":
            # ONE call: the MAC network first, on a stream of its own inside the library, the commitments in the form of their
            # kernel that leaves register room for it (the two parts' scalars / commitments are contiguous: alloc())
            icc.kzg_crebuild_stage_device(d_rows.data_ptr(), n, args.write_step, bufs[0][0].data_ptr(), bufs[1][0].data_ptr(),
                                          bufs[0][1].data_ptr(), bufs[0][2].data_ptr(), d_macs.data_ptr(), bufs[0][3].data_ptr(),
                                          bufs[1][3].data_ptr(), stream)
        elif mode == "reference_order":
            run_chain(torch, icc, mx, d_rows, d_macs, n, args.write_step, bufs, stream, reference_order=True)
        else:
            run_chain(torch, icc, mx, d_rows, d_macs, n, args.write_step, bufs, stream, side.cuda_stream if mode == "two_streams" else None)

    def region(mode):
        for _ in range(args.warmup):
            step(mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(mode)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    if os.environ.get("PORLA_CREBUILD_ONLY_STAGE"):      # (the kernel-trace script: only the one-call form, nothing else in the trace)
        el = region("stage_call")
        print(json.dumps({"stage_call_ms_per_step": round(el / args.steps * 1e3, 4)}))
        return
    el_ref = region("reference_order")   # per part: encode, commitments, MAC encode; one stream
    el_one = region("one_stream")        # both MAC halves from one network, one stream
    el_two = region("two_streams")       # ... and on a second stream beside the data side, as three separate calls
    el = region("stage_call")            # porla_kzg_crebuild_stage_device: the whole stage as ONE call (the reported value)
    # per-kernel split: HIP events around every kernel of two more steps
    mx.profile_enable(True)
    for _ in range(2):
        step("one_stream")
    torch.cuda.synchronize()
    prof = mx.profile_get()
    mx.profile_enable(False)
    per_step = {name: round(ms / 2, 4) for name, ms, cnt in prof}
    per_launch = {name: ms / max(cnt, 1) for name, ms, cnt in prof}
    launches = {name: cnt // 2 for name, ms, cnt in prof}

    # ---- a 2^10-row sample of the same chain against the oracle chain
    cpu = None
    verified = None
    if not args.no_cpu:
        m = min(n, 1 << 10)
        rows = bytes(d_rows[:m * NCOLS * 32].cpu().numpy())
        macs = bytes(d_macs[:64 * m].cpu().numpy())
        sbufs = alloc(torch, m, dev)
        icc.kzg_crebuild_stage_device(d_rows.data_ptr(), m, args.write_step, sbufs[0][0].data_ptr(), sbufs[1][0].data_ptr(),
                                      sbufs[0][1].data_ptr(), sbufs[0][2].data_ptr(), d_macs.data_ptr(), sbufs[0][3].data_ptr(),
                                      sbufs[1][3].data_ptr(), stream)
        torch.cuda.synchronize()
        # the separate calls (porla_icc_encode_xy_device, porla_kzg_commit_batch_device, porla_icc_mac_encode_xy_device) leave the same bytes
        tbufs = alloc(torch, m, dev)
        run_chain(torch, icc, mx, d_rows[:m * NCOLS * 32], d_macs[:64 * m], m, args.write_step, tbufs, stream, side.cuda_stream)
        torch.cuda.synchronize()
        same_as_separate = all(bool((a == b).all().item()) for part in (0, 1) for a, b in zip(sbufs[part], tbufs[part]))
        L = common.oracle()
        L.oracle_kzg_init_key(TAU, ctypes.c_size_t(16), ALPHA, ctypes.c_size_t(16))
        L.oracle_kzg_init_srs(ctypes.c_size_t(NCOLS), (1).to_bytes(32, "big"))
        srs = ctypes.create_string_buffer(64 * NCOLS)
        L.oracle_kzg_srs_g1_raw(srs)
        cores = common.ncpu()
        verified = True
        t1 = time.perf_counter()
        for part in (0, 1):
            x = ctypes.create_string_buffer(64 * m * NCOLS)
            al = ctypes.create_string_buffer(32 * m * NCOLS)
            sc = ctypes.create_string_buffer(32 * m * NCOLS)
            L.oracle_icc_crebuild(rows, ctypes.c_size_t(m), ctypes.c_size_t(NCOLS), 0, part, ctypes.c_uint64(args.write_step), x, al, sc, cores)
            am = common.oracle_commit_batch("bn254", sc.raw, m, NCOLS, srs.raw, threads=cores)
            mh = ctypes.create_string_buffer(64 * m)
            L.oracle_icc_mac_crebuild(macs, ctypes.c_size_t(m), 0, part, ctypes.c_uint64(args.write_step), mh, cores)
            got = [bytes(t.cpu().numpy()) for t in sbufs[part]]
            verified = verified and got[0] == al.raw and got[1] == sc.raw and got[2] == am and got[3] == mh.raw
        cpu_s = time.perf_counter() - t1
        verified = verified and same_as_separate
        cpu = {"value": round(m / cpu_s, 2), "unit": "rows/s", "cores": cores, "kind": "port",
               "sample": "a %d-row chain, both parts: oracle/icc_ref.c encode + alignment scalars, oracle/bn254_ref.c commitment "
                         "per row, oracle/mac_ref.c MAC encode (CPU restatements, not NTL / gnark) over %d threads; %.2f s wall"
                         % (m, cores, cpu_s)}

    # algorithmic HBM bytes per row and step: 128 chunks in; per part 128 x (32 aligned + 32 scalar) out, the scalars read again by
    # the commitment (+ 64 out), the MAC in and out
    bytes_per_row = NCOLS * 32 + 2 * (NCOLS * 64 + NCOLS * 32 + 64 + 64 + 64)
    dom = max(per_step, key=per_step.get) if per_step else None
    rl = None
    if dom:
        rl = {"bound": "hbm", "kernel": dom, "achieved": round(bytes_per_row * n / (sum(per_step.values()) * 1e-3) / 1e9, 3),
              "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None,
              "note": "whole chain: algorithmic bytes of a step / the sum of its kernels' HIP-event times; every kernel of the chain is "
                      "bound by VALU integer issue or by dependent-addition latency (DESIGN.md s4), not by HBM",
              "kernel_ms": round(per_launch[dom], 4)}
        rl["frac"] = round(rl["achieved"] / HBM_PEAK_GBPS, 6)
    out = {"metric": "CRebuild last stage, rows/s (2^%d rows x 128 chunks, X and Y parts: encode + align_MAC + MAC encode)" % args.log2rows,
           "value": round(n * args.steps / el, 1), "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32x9 residue pair (ICC) / u32x8 256-bit modular integer (commitments, MACs)", "data": "synthetic",
           "config": {"workload": "Server::CRebuild_Cached last stage, device-resident: per part porla_icc_encode_device -> "
                                  "porla_kzg_commit_batch_device (align_MAC), and porla_icc_mac_encode_xy_device (both MAC halves from one network), no host sync",
                      "rows": n, "columns": NCOLS, "write_step": args.write_step, "commitments_per_row": 2,
                      "streams": "2 inside the call: the MAC network (started first) beside the data side (encode, commitments in their "
                                 "two-waves-per-SIMD form), no host sync"},
           "entry_point": "porla_kzg_crebuild_stage_device (one call per step)",
           "separate_calls_two_streams_ms_per_step": round(el_two / args.steps * 1e3, 4),
           "one_stream_ms_per_step": round(el_one / args.steps * 1e3, 4), "one_stream_rows_per_s": round(n * args.steps / el_one, 1),
           "reference_order_ms_per_step": round(el_ref / args.steps * 1e3, 4),
           "reference_order_rows_per_s": round(n * args.steps / el_ref, 1),
           "kernels_ms_per_step": per_step, "kernel_launches_per_step": launches,
           "sum_kernels_ms_per_step": round(sum(per_step.values()), 4), "hbm_bytes_per_row": bytes_per_row,
           "roofline": rl, "cpu_baseline": cpu, "bit_exact_vs_oracle": verified}
    print(json.dumps(out))
    if verified is False:
        print("ERROR: GPU chain differs from the oracle chain", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
