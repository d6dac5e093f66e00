"""CPU: the GLV scalar split compiled into the engine (porla_amd/csrc/glv.hip.h, executed on the host through the diagnostic
entry porla_glv_split) against the bit-for-bit Python model of tools/gen_glv.py, and the defining identity
k = k1 + lambda * k2 (mod n) with |k1|, |k2| below the bound the window count relies on.  The reference's secp256k1 path
performs the same split (secp256k1_scalar_split_lambda, porla/Utils/secp256k1_lib/scalar_impl.h:123-156)."""
import ctypes
import os
import random
import sys

import pytest

from tests import common

sys.path.insert(0, os.path.join(common.ROOT, "tools"))


@pytest.mark.parametrize("curve,name", [(0, "Bn254"), (1, "Secp256k1")])
def test_split_matches_model_and_identity(curve, name):
    import gen_glv
    from porla_amd import lib
    d = gen_glv.derive(name, gen_glv.CURVES[name])
    n, lam = d["n"], d["lam"]
    bits = max(((abs(d["a1"]) + abs(d["a2"])) // 2 + 2).bit_length(), ((abs(d["b1"]) + abs(d["b2"])) // 2 + 2).bit_length())
    rnd = random.Random(11)
    ks = [0, 1, 2, n - 1, n - 2, lam, lam + 1, n - lam, lam * lam % n, (n - 1) // 2, (n + 1) // 2, (1 << 128) - 1, 1 << 128,
          abs(d["a1"]), abs(d["b1"]), abs(d["a2"]), abs(d["b2"]), n, n + 1, (1 << 256) - 1]
    ks += [rnd.randrange(1 << 256) for _ in range(20000)]
    m1, m2 = ctypes.create_string_buffer(16), ctypes.create_string_buffer(16)
    n1, n2 = ctypes.c_int(0), ctypes.c_int(0)
    for k in ks:
        assert lib.porla_glv_split(curve, k.to_bytes(32, "big"), m1, ctypes.byref(n1), m2, ctypes.byref(n2)) == 0
        got = [(int.from_bytes(m1.raw, "big"), n1.value), (int.from_bytes(m2.raw, "big"), n2.value)]
        assert got == gen_glv.split_model(d, k % n), hex(k)
        k1 = -got[0][0] if got[0][1] else got[0][0]
        k2 = -got[1][0] if got[1][1] else got[1][0]
        assert (k1 + lam * k2 - k) % n == 0
        assert got[0][0] < 1 << bits and got[1][0] < 1 << bits
