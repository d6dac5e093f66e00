"""GPU: the hand-scheduled gfx950 Montgomery product (porla_amd/csrc/fe_mul_gfx950.inc) against the portable C++ product the
host pass uses, for every field of the engine: random / unreduced / all-ones-limb operands, operands that are partly
compile-time constants, and the chained wide reduction of the audit path (tools/fe_check.hip)."""
import os
import subprocess

import pytest

from tests import common

pytestmark = pytest.mark.gpu


def test_assembly_product_matches_portable_product():
    exe = os.path.join(common.ROOT, "porla_amd", "fe_check")
    assert os.path.exists(exe), "build it with make -C porla_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK")
