"""GPU box: every behaviour switch the library still reads from the environment (DESIGN.md s8 lists them) gives the oracle's
bytes.  A switch is read once per process, so each case is a child pytest:
  PORLA_MSM_SHARED_BUCKETS=0  one complete MSM per host range instead of ranges accumulating into one bucket set
  PORLA_MSM_SMALL=0           the general bucket path at audit sizes (instead of the single-launch MSM)
  PORLA_MAC_QUAD_MAX=0        one lane per MAC butterfly / per mix element / per scaled MAC: the kernels N > 2^16 rows take
  PORLA_MAC_QUAD_MAX=5        the boundary inside the tests' sizes (both families in one encode's neighbourhood)
  PORLA_COMMIT_SMALL=0        batch kernels for single rows (instead of the single-launch commitment)
  PORLA_COMMIT_TABLE_GB=1     a table budget that forces narrow windows (the shrink loop of FixedBase::build)
  PORLA_COMMIT_WINDOW=11      a fixed window width
  PORLA_NO_ADX=1              portable host field products
The operational ones (PORLA_MSM_DEVICES, PORLA_MSM_SPLIT_MIN: tests/test_msm_multi_gpu.py; PORLA_RCCL_LIB,
PORLA_DIST_INIT_TIMEOUT_S: tests/test_sharded_gloo.py / the two-rank bench tests) are covered where they act."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "PORLA_MSM_SHARED_BUCKETS=0": "tests/test_msm_multi_gpu.py -k 'forced_shards or degenerate'",
    "PORLA_MSM_SMALL=0": "tests/test_msm_small_gpu.py -k 'audit or shape or sizes'",
    "PORLA_MAC_QUAD_MAX=0": "tests/test_mac_fft_gpu.py tests/test_mix_gpu.py",
    "PORLA_MAC_QUAD_MAX=5": "tests/test_mac_fft_gpu.py",
    "PORLA_COMMIT_SMALL=0": "tests/test_fixed_base_gpu.py -k 'small or single or coalesc or row'",
    "PORLA_COMMIT_TABLE_GB=1": "tests/test_fixed_base_gpu.py tests/test_golden_gpu.py -k 'commit or kzg or digest'",
    "PORLA_COMMIT_WINDOW=11": "tests/test_fixed_base_gpu.py -k 'commit or batch'",
    "PORLA_NO_ADX=1": "tests/test_msm_bn254_gpu.py -k 'edge or audit_like or kat or eip'",
}


@pytest.mark.parametrize("switch", sorted(CASES))
def test_switch(switch):
    import shlex
    name, value = switch.split("=")
    env = dict(os.environ)
    env[name] = value
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + shlex.split(CASES[switch])
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout[-1500:] + r.stderr[-1500:])
    assert r.returncode in (0, 5), tail          # 5 = no test matched the -k expression: treated as a failure below
    assert r.returncode == 0 and " passed" in r.stdout, tail
