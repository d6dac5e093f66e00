#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM throughput of the MI355X engine (BASELINE.json metric), one rank per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: ONE 2^20-pair MSM per GPU (BASELINE.json config 2,
"KZG scheme, single 2^20-point BN254 G1 MSM on 1 MI355X"), inputs resident in HBM in the reference's wire
format (32-B big-endian scalars + 64-B X||Y points, porla/main.go:118-138) when the timed region starts.
With N > 1 every rank owns its own 2^20 pairs (input-range sharding, weak scaling), produces one partial
Jacobian sum, the 96-byte partials are exchanged with one RCCL all_gather and folded with N-1 group additions
 -- the whole job is one N*2^20-pair MSM per step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_bucket_sum), timed with HIP events on the
launch stream inside the library over the timed region; `cpu_baseline` is the oracle (oracle/bn254_ref.c, a
CPU restatement -- NOT gnark) on the same inputs, which also serves as the bit-exactness check of this run.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_PAIR = 96      # 32-B scalar + 64-B affine point


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=20, help="pairs per GPU = 2^log2n (default: the 2^20 of BASELINE.json)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / bit-exact check leg")
    ap.add_argument("--traffic", type=float, default=None, help="HBM bytes per launch from a separate rocprofv3 --pmc run")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from porla_amd import multiexp as mx
    from porla_amd import lib
    from tests import common  # oracle access is allowed here for input generation + the cpu_baseline leg only

    n = 1 << args.log2n
    # ---- synthetic inputs (SURVEY.md s8d cfg 2): points k_i*G, scalars SHA-256 stream (81 % of them >= r)
    t0 = time.time()
    if rank == 0:
        common.cached_inputs(n)          # rank 0 generates (or finds) the cache; the others read it
    if world > 1:
        dist.barrier()
    sc0, pt = common.cached_inputs(n)
    sc = sc0 if rank == 0 else common.synth_scalars(n, start=rank * n)  # every rank: its own scalars, same base
    gen_s = time.time() - t0
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    from porla_amd import sharded

    def step():
        # N == 1: one MSM -> 64-byte affine.  N > 1: per-rank partial Jacobian, ONE RCCL all_gather of N x 96 bytes,
        # N-1 group additions + one inversion on the host (porla_amd/sharded.py)
        return sharded.sharded_msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, stream, dev)

    result = None
    for _ in range(args.warmup):
        result = step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    mx.profile_enable(True)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    sync()
    elapsed = time.perf_counter() - t_start
    prof = mx.profile_get()
    mx.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed / 1e6
        kern = {name: (ms / max(cnt, 1)) for name, ms, cnt in prof}
        dom = max(kern, key=kern.get) if kern else None
        roofline = None
        if dom:
            achieved = ALGO_BYTES_PER_PAIR * n / (kern[dom] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                        "traffic": args.traffic, "kernel_ms": round(kern[dom], 4),
                        "all_kernels_ms": {k: round(v, 4) for k, v in kern.items()}}
        cpu = None
        verified = None
        if not args.no_cpu and world == 1:
            cores = common.ncpu()
            t1 = time.perf_counter()
            want = common.oracle_msm(sc, pt, n, threads=cores)
            cpu_s = time.perf_counter() - t1
            verified = (want == result)
            cpu = {"value": round(n / cpu_s / 1e6, 4), "unit": "Mmul/s", "cores": cores, "kind": "port",
                   "sample": "the same 2^%d pairs, oracle/bn254_ref.c bucket MSM range-split over %d threads "
                             "(CPU restatement, not gnark); %.1f s wall" % (args.log2n, cores, cpu_s)}
        out = {
            "metric": "BN254 G1 MSM Mscalar-mul/s at 2^20 pts", "value": round(value, 3), "unit": "Mmul/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (256-bit modular integer)",
            "data": "synthetic",
            "config": {"workload": "KZG scheme, single 2^%d-point BN254 G1 MSM per GPU, inputs resident in HBM, "
                                   "output 64-B affine point" % args.log2n,
                       "pairs_per_gpu": n, "sharding": "input-pair range per rank + RCCL all_gather of 96-B Jacobian partials"
                       if world > 1 else "single GPU", "input_gen_s": round(gen_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "bit_exact_vs_oracle": verified,
            "result": result.hex() if result else None,
        }
        print(json.dumps(out))
        if verified is False:
            print("ERROR: GPU result differs from the oracle", file=sys.stderr)
            sys.exit(1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
