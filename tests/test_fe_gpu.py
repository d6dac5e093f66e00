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


def test_reduced_radix_product_matches_portable_product():
    """fe30.hip.h (9 x 30-bit limbs, radix 2^270; the bucket accumulation's field form): the generated assembly blocks equal
    the portable form limb for limb and the 8 x 32-bit Montgomery product after the radix change, for reduced, unreduced
    (< 2^258), all-ones, p and 1 operands, products and squares, over every modulus (tools/fe30_check.hip)"""
    exe = os.path.join(common.ROOT, "porla_amd", "fe30_check")
    assert os.path.exists(exe), "build it with make -C porla_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 mismatches") == 4, r.stdout
    # and the division-step inversion of the finish kernels (inv30.hip.h): a * a^-1 = 1 over 4096 residues per base field
    assert r.stdout.count(" 0 wrong inverses") == 2, r.stdout
