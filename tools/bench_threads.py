#!/usr/bin/env python3
"""Aggregate rate of blocking device-resident BN254 MSMs of 2^log2n pairs issued from T host threads at once (each thread its own
stream; the engine leases every concurrent blocking call its own workspace slot): is a mid-size MSM bound by the host's launch rate?
Usage: bench_threads.py [log2n] [threads ...]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import multiexp as mx
from tests import common
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
threads = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 6, 8]
n = 1 << lg
sc, pt = common.cached_inputs(1 << 20)
d_sc = torch.frombuffer(bytearray(sc[:32 * n]), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt[:64 * n]), dtype=torch.uint8).cuda()
want = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
reps = int(os.environ.get("REPS", "200"))
for T in threads:
    streams = [torch.cuda.Stream() for _ in range(T)]
    bad = []
    def work(k):
        s = streams[k].cuda_stream
        for _ in range(reps):
            if mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s) != want:
                bad.append(k)
    for rnd in range(2):
        th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        el = time.perf_counter() - t0
    assert not bad
    print("2^%d pairs, %d threads: %.4f ms per MSM (%.1f Mmul/s)" % (lg, T, el / (T * reps) * 1e3, n * T * reps / el / 1e6), flush=True)
