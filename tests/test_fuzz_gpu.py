"""GPU: a short run of the randomised differential test (tools/fuzz_msm.py: sizes x window overrides 2..20 x GLV on/off x
scalar distributions x repeated / infinity / cancelling points, both curves) against the oracle.  The long runs are recorded
in profiles/ (98 641 cases, 0 mismatches at the time of writing)."""
import os
import subprocess
import sys

import pytest

from tests import common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_short_fuzz(seed):
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "tools", "fuzz_msm.py"), "8", str(seed)],
                       capture_output=True, text=True, timeout=300, cwd=common.ROOT)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("tool,seed", [("fuzz_commit.py", 21), ("fuzz_icc.py", 22), ("fuzz_audit.py", 23)])
def test_short_fuzz_of_commitments_and_icc(tool, seed):
    """the same for the fixed-base commitments (tools/fuzz_commit.py: table windows, 1 .. 3000 rows, short rows, padded strides,
    infinity base points, both curves -- single-launch path, slice fold, host and device normalisation) and for the ICC kernels
    (tools/fuzz_icc.py: data encode, mix, MAC encode, MAC mix) and the audit side (tools/fuzz_audit.py: row combine, the gathered pair of
    MSMs in both forms, the digest batch); long runs: profiles/r02_y_fuzz_late_build.txt, r02_zd_*, r03_l_*, r03_s_*"""
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "tools", tool), "8", str(seed)],
                       capture_output=True, text=True, timeout=300, cwd=common.ROOT)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
