"""The plain-C consumer of the reference's plug-in boundary (integration/kzg_harness/harness.c): compiled with gcc against
include/libmultiexp.h and linked with -lmultiexp exactly as porla/Makefile:13 links the Go-built library."""
import os
import subprocess

import pytest

from tests import common

HARNESS_DIR = os.path.join(common.ROOT, "integration", "kzg_harness")


def build():
    subprocess.check_call(["make", "-C", HARNESS_DIR], stdout=subprocess.DEVNULL)
    return os.path.join(HARNESS_DIR, "harness")


def test_client_side_calls_need_no_gpu():
    r = subprocess.run([build(), "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "HARNESS OK" in r.stdout, r.stdout + r.stderr
    assert "FAIL" not in r.stdout


@pytest.mark.gpu
def test_full_kzg_call_sequence_on_gpu():
    r = subprocess.run([build(), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HARNESS OK" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("ok:") >= 11
