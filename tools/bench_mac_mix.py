#!/usr/bin/env python3
"""MAC part of Server::mix (porla_icc_mac_mix_host): wall time of the host call and the kernel's own time by length."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from porla_amd import icc, multiexp as mx, lib
from tests import common
import ctypes
for lg in (0, 4, 8, 10, 12, 14):
    ln = 1 << lg
    n_total = 1 << 15
    pts = common.synth_points(min(2 * ln, 4096), start=100)
    pts = (pts * (2 * ln // min(2 * ln, 4096) + 1))[:64 * 2 * ln]
    a0, a1 = pts[:64 * ln], pts[64 * ln:]
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = icc.mac_mix_host(a0, a1, ln, n_total, "bn254") if hasattr(icc, "mac_mix_host") else None
    wall = (time.perf_counter() - t0) / reps * 1e3
    mx.profile_enable(True)
    icc.mac_mix_host(a0, a1, ln, n_total, "bn254")
    prof = {k: round(ms / max(c, 1), 3) for k, ms, c in mx.profile_get()}
    mx.profile_enable(False)
    print(json.dumps({"len": ln, "host_call_ms": round(wall, 3), "kernels_ms": prof}), flush=True)
