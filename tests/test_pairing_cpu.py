"""CPU: the host pairing behind verify_proof (porla/main.go:177-193 -> kzg.Verify).  Bilinearity through the diagnostic entry
points, and the fast form (projective Miller steps, easy/hard final exponentiation) against the literal reference form
(affine steps, exponent (p^12 - 1)/r).  The hard part's addition-chain exponent is checked symbolically."""
import ctypes
import time

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
X = 4965661367192848881


def test_hard_part_exponent_identity():
    """y0 * y1^2 * y2^6 * y3^12 * y4^18 * y5^30 * y6^36 of f12_final_exp has exponent (p^4 - p^2 + 1) / r exactly"""
    assert P == 36 * X**4 + 36 * X**3 + 24 * X**2 + 6 * X + 1 and R == 36 * X**4 + 36 * X**3 + 18 * X**2 + 6 * X + 1
    e = (P + P**2 + P**3) - 2 + 6 * X * X * P * P - 12 * X * P - 18 * (X + X * X * P) - 30 * X * X - 36 * (X**3 + X**3 * P)
    assert e * R == P**4 - P**2 + 1


def test_bilinearity_fast_and_reference_forms_agree():
    from porla_amd import lib, multiexp as mx
    G1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")

    def g2(k):
        out = ctypes.create_string_buffer(128)
        assert lib.porla_bn254_g2_mul_generator((k % R).to_bytes(32, "big"), out) == 0
        return out.raw

    def g1(k):
        return mx.bn254_mult(G1, (k % R).to_bytes(32, "big"))

    a, b = 0x1234567890abcdef1234567890abcdef, 0xfedcba0987654321fedcba09
    cases = [(g1(a), g2(b), g1(R - a * b % R), g2(1), 1),          # e(aG, bH) e(-abG, H) = 1
             (g1(a), g2(b), g1(R - (a * b + 1) % R), g2(1), 0),    # off by one
             (g1(a * b), g2(1), g1(R - a), g2(b), 1),
             (g1(5), g2(7), g1(R - 7), g2(5), 1),
             (g1(5), g2(7), g1(R - 7), g2(6), 0),
             (bytes(64), g2(3), bytes(64), g2(4), 1)]              # infinity on the G1 side: both pairings are 1
    for i, (p1, q1, p2, q2, want) in enumerate(cases):
        fast = lib.porla_bn254_pairing_product_is_one(p1, q1, p2, q2, 0)
        assert fast == want, i
        if i < 3:     # the literal form costs ~40 ms a piece
            assert lib.porla_bn254_pairing_product_is_one(p1, q1, p2, q2, 1) == want, i
    t0 = time.perf_counter()
    for _ in range(5):
        lib.porla_bn254_pairing_product_is_one(*cases[0][:4], 0)
    assert (time.perf_counter() - t0) / 5 < 0.05
