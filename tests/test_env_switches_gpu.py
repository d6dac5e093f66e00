"""GPU box: the documented off-switches still give the oracle's bytes.  Every switch is read once per process, so each case is a
child process: PORLA_MSM_SHARED_BUCKETS=0 (one complete MSM per host range), PORLA_MSM_SMALL=0 (general path at audit sizes),
PORLA_ICC_F30=0 (8 x 32-bit ICC kernel), PORLA_ICC_SPLIT=0 (the reduced-radix kernel with both residues side by side in LDS:
icc30.hip.h), PORLA_MAC_QUAD=0 (one lane per MAC butterfly), PORLA_ICC_MIX30=0 (Server::mix's data part in the 2^256 field form), PORLA_KZG_EVAL30=0 (the digest batch's Horner
evaluation with one lane per row), PORLA_KZG_EVAL_LAZY=0 (Horner with eight lanes per row instead of the dot product with the
reduction at the end), PORLA_COMMIT_SMALL=0 (batch kernels for
single rows), PORLA_NO_ADX=1 (portable host field products), PORLA_TREE_SPLIT=1 (reduction tree on one stream),
PORLA_FRONT_SPLIT=0 (point conversion on the MSM's own stream)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "PORLA_MSM_SHARED_BUCKETS=0": "tests/test_msm_multi_gpu.py -k 'forced_shards or degenerate'",
    "PORLA_MSM_SMALL=0": "tests/test_msm_small_gpu.py -k 'audit or shape or sizes'",
    "PORLA_ICC_F30=0": "tests/test_icc_gpu.py -k 'not full'",
    "PORLA_ICC_SPLIT=0": "tests/test_icc_gpu.py tests/test_golden_gpu.py -k 'icc or crebuild or edge or three_passes'",
    "PORLA_MAC_QUAD=0": "tests/test_mac_fft_gpu.py tests/test_mix_gpu.py",
    "PORLA_ICC_MIX30=0": "tests/test_mix_gpu.py tests/test_hadd_gpu.py",
    "PORLA_KZG_EVAL30=0": "tests/test_fixed_base_gpu.py tests/test_golden_gpu.py -k 'digest or kzg'",
    "PORLA_KZG_EVAL_LAZY=0": "tests/test_fixed_base_gpu.py tests/test_golden_gpu.py -k 'digest or kzg'",
    "PORLA_COMMIT_SMALL=0": "tests/test_fixed_base_gpu.py -k 'small or single or coalesc or row'",
    "PORLA_NO_ADX=1": "tests/test_msm_bn254_gpu.py -k 'edge or audit_like or kat or eip'",
    "PORLA_TREE_SPLIT=1": "tests/test_msm_bn254_gpu.py -k 'full or 2p20 or uniform'",
    "PORLA_FRONT_SPLIT=0": "tests/test_msm_bn254_gpu.py -k 'full or 2p20 or uniform'",
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
