"""GPU box: the multi-rank path of bench.py end to end with 2 ranks -- range sharding, per-rank partial Jacobian, all_gather
of the 96-byte partials, host fold, MAX-over-ranks timing, whole-job oracle check.  The box has one GPU, so the two ranks
share it and the partials travel over gloo (PORLA_DIST_BACKEND=gloo); the driver's 8-GPU runs use nccl = RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests import common

pytestmark = pytest.mark.gpu


def test_two_rank_bench_line_is_bit_exact():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PORLA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-commits"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=common.ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["bit_exact_vs_oracle"] is True          # the 2 * 2^20-pair whole-job MSM against the oracle
    assert d["cpu_baseline"] is None                 # reported at N = 1 only
    assert d["roofline"]["kernel"] and d["value"] > 0
    assert 0 < d["roofline"]["int_multiplier"]["frac"] <= 1.0


def test_single_rank_bench_line_prices_the_kernel_it_timed():
    """N = 1, the default workload with its audit-size and host-boundary legs: one JSON line, the roofline's two fractions in
    (0, 1] -- the multiplication count must be the timed 2^20-pair MSM's, not that of an audit-size MSM run after it --
    and every leg bit-exact"""
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-commits"],
                       capture_output=True, text=True, timeout=600, cwd=common.ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["bit_exact_vs_oracle"] is True and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["kernel"] == "k_bucket_sum30" and 0 < rf["frac"] <= 1.0 and 0.3 < rf["int_multiplier"]["frac"] <= 1.0
    assert abs(rf["achieved"] - 96 * (1 << 20) / (rf["kernel_ms"] * 1e-3) / 1e9) < 0.01 * rf["achieved"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["host_boundary"]["same_result"] is True
    for group in d["audit_size_msm"].values():
        if isinstance(group, dict):
            assert all(v["bit_exact_vs_oracle"] for v in group.values())
