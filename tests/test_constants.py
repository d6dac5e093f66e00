"""CPU: every hard-coded numeric table in the engine sources equals the value derived from the modulus alone."""
import os
import re
import sys

from tests import common

sys.path.insert(0, os.path.join(common.ROOT, "tools"))
import gen_constants as gc  # noqa: E402

CSRC = os.path.join(common.ROOT, "porla_amd", "csrc")


def struct_tables(path, struct):
    text = open(path).read()
    m = re.search(r"struct %s\s*\{(.*?)\n\};" % struct, text, flags=re.S)
    assert m, struct
    body = m.group(1)
    out = {}
    for name in ("P", "R1", "R2", "ORDER", "R1_30", "R2_30"):
        mm = re.search(r"\b%s\[8\]\s*=\s*\{([^}]*)\}" % name, body)
        if mm:
            out[name] = [int(x.strip().rstrip("u"), 0) for x in mm.group(1).split(",") if x.strip()]
    mm = re.search(r"\bINV\s*=\s*(0x[0-9a-fA-F]+)u", body)
    if mm:
        out["INV"] = int(mm.group(1), 16)
    return out


def test_field_parameter_packs():
    for path, struct, p in ((os.path.join(CSRC, "fe.hip.h"), "Bn254Fp", gc.BN_P),
                            (os.path.join(CSRC, "fe.hip.h"), "Secp256k1Fp", gc.SECP_P),
                            (os.path.join(CSRC, "host_curve.hpp"), "Bn254Fr", gc.BN_R)):
        t = struct_tables(path, struct)
        m = gc.mont(p)
        assert t["P"] == m["P"] and t["INV"] == m["INV"], struct
        if struct == "Secp256k1Fp":
            # special-form field: plain residues (radix 1), product folded with 2^256 = 2^32 + 977 (mod p)
            assert t["R1"] == [1] + [0] * 7 and t["R2"] == [1] + [0] * 7
            assert p == 2**256 - 2**32 - 977
        else:
            assert t["R1"] == m["R1"] and t["R2"] == m["R2"], struct
        if struct == "Bn254Fp":   # reduced-radix form of fe30.hip.h: Montgomery radix 2^270
            assert t["R1_30"] == gc.limbs(pow(2, 270, p)) and t["R2_30"] == gc.limbs(pow(2, 256 + 270, p))


def test_group_orders():
    assert struct_tables(os.path.join(CSRC, "msm.hip.h"), "Bn254G1")["ORDER"] == gc.limbs(gc.BN_R)
    assert struct_tables(os.path.join(CSRC, "msm.hip.h"), "Secp256k1G")["ORDER"] == gc.limbs(gc.SECP_N)


def test_final_exponent_and_ate_loop():
    assert open(os.path.join(CSRC, "final_exp_limbs.inc")).read() == gc.final_exp_inc()
    text = open(os.path.join(CSRC, "pairing_host.hpp")).read()
    m = re.search(r"S\[4\]\s*=\s*\{([^}]*)\}", text)
    assert [int(x.strip().rstrip("u"), 0) for x in m.group(1).split(",")] == gc.limbs(6 * gc.BN_X ** 2, 4)


def test_reduced_radix_column_sums_fit_64_bits():
    """fe30.hip.h accumulates the 16..18 products of a column in one 64-bit register without carries: the worst case must fit"""
    import check_fe30_bounds as cb
    assert all(cb.check(name, p) for name, p in cb.MODULI.items())
    # the two-product forms (one reduction / one fold for a b + c d): their side-chain columns are the ones the generator uses
    import gen_fe30_asm as g30
    assert cb.MUL2_CHAIN == g30.MUL2_CHAIN and cb.PM_MUL2_CHAIN == g30.PM_MUL2_CHAIN[0] == g30.PM_MUL2_CHAIN[1]
    assert all(cb.check_mul2(name, p) for name, p in cb.MODULI.items() if name != "p_icc") and cb.check_pm_mul2()
    assert not cb.check_mul2("no chain", cb.MODULI["bn254_p"], chain=(99, 99))       # ... and they are needed
    # the operand ranges the group law leans on (2 Y <= 8 p in a doubling, X <= 5 p, Y <= 4 p in the lazy memory form) against the
    # products' budgets, and the borrow-free subtraction tables incl. the doubled one of f30_sub_twice (advisor r4)
    assert cb.check_value_ranges() and cb.check_sub_tables()
    src = open(os.path.join(CSRC, "fe30.hip.h")).read()
    assert "F30_MUL2_CHAIN_LO = %d, F30_MUL2_CHAIN_HI = %d" % cb.MUL2_CHAIN in src and "F30_PM_MUL2_CHAIN = %d" % cb.PM_MUL2_CHAIN in src
    # and the generated assembly is what the generator produces
    assert open(os.path.join(CSRC, "fe30_mul_gfx950.inc")).read() == g30.whole()
