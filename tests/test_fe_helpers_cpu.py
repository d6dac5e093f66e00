"""The 8 x 32-bit field helpers the kernels are built from (porla_amd/csrc/fe.hip.h: borrow / carry chains, single-chain negation, the
bounded reduction of the ICC finish step with its quotient estimate), executed on the HOST through porla_diag_fe_op -- the same source
as the device code, the portable branch of the carry primitives -- against Python integers.  The device branch (the compiler's carry
builtins) is covered by every GPU parity test; what this pins without a GPU is the arithmetic itself, in particular the quotient
estimate's claim qe <= floor(t / q) <= qe + 1 at the multiples of q where it is tight."""
import ctypes
import random

import pytest

from porla_amd.loader import load

R_BN254 = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001        # BN254 group order (main.go: fr)
N_SECP = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141          # secp256k1 group order
P_ICC = 207 * (1 << 248) + 1                                                          # porla/Utils/utils.h:31-32
MODS = {0: R_BN254, 1: N_SECP}


def _op(lib, op, modulus, a, b=None):
    out = ctypes.create_string_buffer(32)
    rc = lib.porla_diag_fe_op(op, modulus, a.to_bytes(32, "little"), None if b is None else b.to_bytes(32, "little"), out)
    assert rc == 0
    return int.from_bytes(out.raw, "little")


@pytest.fixture(scope="module")
def lib():
    return load()


def test_bounded_reduction_at_every_multiple_of_the_modulus(lib):
    q = R_BN254
    cases = [0, 1, P_ICC - 1, P_ICC - 2, (1 << 255), (1 << 255) - 1]
    for k in range(0, 5):
        for d in (-2, -1, 0, 1, 2):
            cases.append(k * q + d)
    # values whose top word sits at the edges of the estimate's quotient buckets: t7 = m (q7 + 1) - 1, m (q7 + 1), with every lower word 0 / all ones
    q7 = q >> 224
    for m in range(1, 5):
        for t7 in (m * (q7 + 1) - 1, m * (q7 + 1), m * q7, m * q7 + 1):
            cases.append(t7 << 224)
            cases.append((t7 << 224) | ((1 << 224) - 1))
    rng = random.Random(20261005)
    cases += [rng.randrange(P_ICC) for _ in range(4000)]
    for t in cases:
        if 0 <= t < 5 * q and t < (1 << 256):
            assert _op(lib, 0, 0, t) == t % q, hex(t)
    # the secp256k1 order: one conditional subtraction (the stream's values stay below p_icc < n there)
    n = N_SECP
    for t in [0, 1, n - 1, n, n + 1, (1 << 256) - 1, P_ICC - 1] + [rng.randrange(1 << 256) for _ in range(2000)]:
        assert _op(lib, 0, 1, t) == (t - n if t >= n else t), hex(t)


def test_negation_and_the_two_operand_chains(lib):
    rng = random.Random(7)
    for modulus, q in MODS.items():
        vals = [0, 1, 2, q - 1, q - 2, (q + 1) // 2, (1 << 224), (1 << 32) - 1, (1 << 32), ((1 << 64) - 1) << 96] + [rng.randrange(q) for _ in range(2000)]
        for a in vals:
            assert _op(lib, 1, modulus, a) == (-a) % q
            assert _op(lib, 2, modulus, a) == (-a) % q
        for _ in range(2000):
            a, b = rng.randrange(q), rng.randrange(q)
            assert _op(lib, 3, modulus, a, b) == (a - b) % q
            assert _op(lib, 4, modulus, a, b) == (a + b) % q
        for a, b in ((0, 0), (0, q - 1), (q - 1, 0), (q - 1, q - 1), (1, q - 1), (q - 1, 1)):
            assert _op(lib, 3, modulus, a, b) == (a - b) % q
            assert _op(lib, 4, modulus, a, b) == (a + b) % q


def test_negation_matches_the_two_chain_form_beyond_the_modulus(lib):
    """fe_neg replaced `0 - a, then + P where that borrowed` by ONE chain `P - a` (zero kept): the same 256-bit word pattern for EVERY
    input, also the non-canonical ones a caller never sends -- the claim the device code's comment makes"""
    rng = random.Random(11)
    for modulus, q in MODS.items():
        for a in [q, q + 1, (1 << 256) - 1, (1 << 255)] + [rng.randrange(1 << 256) for _ in range(1000)]:
            want = 0 if a == 0 else (q - a) % (1 << 256)
            assert _op(lib, 1, modulus, a) == want
