"""bench.py's own N-rank launcher, the parts that need no GPU: `python bench.py --gpus N` with WORLD_SIZE unset
must become a launcher BEFORE anything touches a device, refuse a node with fewer devices than ranks (RCCL: one device per rank),
and hand a failing rank's exit code up instead of hanging."""
import os
import subprocess
import sys
import time

import pytest
import torch

from tests import common

BENCH = os.path.join(common.ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PORLA_DIST_BACKEND")}
    env.update(kw)
    return env


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a node with fewer than 2 GPUs")
def test_more_ranks_than_devices_is_refused_at_once():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=_env(), cwd=common.ROOT)
    assert r.returncode == 2 and "one device per rank" in r.stderr, r.stderr[-1000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]       # no line that could be taken for a measurement
    assert time.time() - t0 < 120


@pytest.mark.skipif(torch.cuda.is_available(), reason="the failing-rank case: a container without a GPU")
def test_failing_ranks_end_the_launch_with_their_exit_code():
    """two ranks over gloo in a container without a GPU: every rank fails (there is no device and NO CPU fallback), torchrun ends
    the group, the launcher returns non-zero and prints no JSON line"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-legs", "--no-commits"],
                       capture_output=True, text=True, timeout=600, env=_env(PORLA_DIST_BACKEND="gloo", PORLA_DIST_TIMEOUT_S="60"),
                       cwd=common.ROOT)
    assert r.returncode != 0, r.stdout[-1000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_the_launcher_runs_before_any_device_call():
    """static check of the order in main(): the self-launch happens right after argument parsing, ahead of the first torch.cuda /
    library call of the process (a process that has initialised the GPU must not be replaced, and the launcher never needs one)"""
    src = open(BENCH).read()
    body = src[src.index("def main():"):]
    launch = body.index("launch_ranks(args.gpus")
    for needle in ("measure_fe_mul_peak() if", "torch.cuda.set_device", "from porla_amd import multiexp"):
        assert launch < body.index(needle), needle
    lr = src[src.index("def launch_ranks"):src.index("def main():")]
    assert "subprocess.Popen" in lr and "os.exec" not in lr and "torch.cuda.is_available" not in lr
