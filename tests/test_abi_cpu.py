"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; the host-side (latency-bound)
entry points of the plug-in -- which legitimately run on the CPU in the product too -- match the golden vectors.
No device compute is invoked here."""
import json
import os
import re

import pytest

from tests import common

ROOT = common.ROOT
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn254_golden.json")))


def declared_symbols():
    syms = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^\s*(?:extern\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            syms.append(m.group(1))
    return sorted(set(syms))


def test_every_declared_symbol_is_exported():
    from porla_amd import lib
    syms = declared_symbols()
    # the 14 cgo symbols of the reference (porla/Utils/libmultiexp.h:71-84) must all be there
    ref14 = ["init_key", "init_SRS", "init_SRS_from_data", "compute_digest", "compute_digest_complement",
             "compute_digest_from_srs", "compute_multi_exp", "compare_commitment", "create_proof", "verify_proof",
             "add_point", "mult_point", "neg_point", "set_inf_point"]
    for s in ref14:
        assert s in syms
    assert len(syms) >= 14 + 8
    for s in syms:
        assert getattr(lib, s) is not None, s


def test_point_ops_match_golden():
    from porla_amd import multiexp as mx
    G = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    for e in GOLD["scalar_mul_G"]:
        assert mx.bn254_mult(G, bytes.fromhex(e["k"])).hex() == e["P"]
    for e in GOLD["point_ops"]:
        a, b = bytes.fromhex(e["a"]), bytes.fromhex(e["b"])
        assert mx.bn254_add(a, b).hex() == e["add"]
        assert mx.bn254_neg(a).hex() == e["neg_a"]
        assert mx.bn254_compare(a, a)
        if e["add"] != e["a"]:
            assert not mx.bn254_compare(a, bytes.fromhex(e["add"]))
    for e in GOLD["mult_point"]:
        assert mx.bn254_mult(bytes.fromhex(e["a"]), bytes.fromhex(e["s"])).hex() == e["r"]
    assert mx.bn254_set_infinity() == bytes(64)
    assert mx.bn254_scalar_set_int(0x7fffffff) == bytes(28) + bytes.fromhex("7fffffff")


def test_kzg_host_side_matches_golden():
    """init_key / init_SRS (wire blob, G1 part) / compute_digest / verify_proof: the client-side calls
    (Client.hpp:159-167,348-354,411-419,1637-1662) need no GPU."""
    from porla_amd import multiexp as mx
    kz = GOLD["kzg"]
    tau, alpha, n = bytes.fromhex(kz["tau"]), bytes.fromhex(kz["alpha"]), kz["n"]
    mx.init_key(tau, alpha)
    blob = mx.init_SRS(n)
    assert len(blob) == 32 * n + 132          # Client.hpp:350-357
    assert blob[:4 + 32 * n].hex() == kz["srs_g1_blob"]
    assert blob.hex() == kz["srs_blob"]       # the WHOLE blob: the two compressed G2 points too (generator, tau * generator)
    for c in kz["cases"]:
        assert mx.compute_digest(bytes.fromhex(c["f"])).hex() == c["digest"]
        for op in c["open"]:
            args = [bytes.fromhex(op[k]) for k in ("commitment", "H", "point", "claim")]
            assert mx.verify_proof(*args)
    op = kz["cases"][0]["open"][0]
    bad_claim = (int(op["claim"], 16) ^ 1).to_bytes(32, "big")
    assert not mx.verify_proof(bytes.fromhex(op["commitment"]), bytes.fromhex(op["H"]), bytes.fromhex(op["point"]), bad_claim)
    assert not mx.verify_proof(bytes.fromhex(op["H"]), bytes.fromhex(op["commitment"]), bytes.fromhex(op["point"]),
                               bytes.fromhex(op["claim"]))
    # server side: SRS from the wire blob (incl. compressed G2), then verify again
    mx.init_SRS_from_data(n, blob)
    assert mx.verify_proof(*[bytes.fromhex(op[k]) for k in ("commitment", "H", "point", "claim")])
    # complement is linear in its scalar (h_MAC is random by design, main.go:52-59)
    s1, s2 = common.synth_scalars(1, 1), common.synth_scalars(1, 2)
    r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    s3 = ((int.from_bytes(s1, "big") + int.from_bytes(s2, "big")) % r).to_bytes(32, "big")
    mx.init_SRS(n)
    assert mx.bn254_add(mx.compute_digest_complement(s1), mx.compute_digest_complement(s2)) == mx.compute_digest_complement(s3)


def test_jac_sum_folds_partials():
    """porla_bn254_jac_sum (the N-GPU fold) on hand-made Jacobian partials"""
    import bn254_py as o
    from porla_amd import multiexp as mx
    pts = [o.g1_mul(o.G1, k) for k in (5, 7, 11)]
    parts = b""
    for i, (x, y) in enumerate(pts):
        z = 3 + i
        parts += (x * z * z % o.P).to_bytes(32, "big") + (y * z ** 3 % o.P).to_bytes(32, "big") + z.to_bytes(32, "big")
    parts += (1).to_bytes(32, "big") + (1).to_bytes(32, "big") + bytes(32)      # infinity
    assert mx.jac_sum("bn254", parts, 4) == o.g1_marshal(o.g1_mul(o.G1, 23))
    assert mx.jac_sum("bn254", b"", 0) == bytes(64)


def test_tree_fold_is_the_weighted_sum_of_its_bit_slices():
    """the MSM's host tail (host_fold64.hpp:h_fold_tree64): windows of S / M_k sums -> sum_w 2^(c w) (S_w + sum_k 2^k M_wk),
    checked against Python integers on G multiples, with empty (infinity) slots"""
    import ctypes
    import random
    import bn254_py as o
    from porla_amd import lib
    rnd = random.Random(5)
    for W, c in ((1, 2), (3, 5), (2, 16), (16, 16), (15, 17)):
        ks, blob = [], b""
        for w in range(W):
            for j in range(c):
                k = 0 if rnd.random() < 0.2 else rnd.randrange(1, 1 << 64)
                ks.append(k)
                blob += o.g1_marshal(o.g1_mul(o.G1, k)) if k else bytes(64)
        want = 0
        for w in range(W):
            v = ks[w * c] + sum(ks[w * c + 1 + k] << k for k in range(c - 1))
            want += v << (c * w)
        out = ctypes.create_string_buffer(64)
        assert lib.porla_bn254_tree_fold(blob, W, c, out) == 0
        assert out.raw == o.g1_marshal(o.g1_mul(o.G1, want % o.R))


def test_eip196_vectors_through_the_host_side_symbols():
    """add_point / mult_point (main.go:195-214) run on the host in the product too: every go-ethereum bn256Add / bn256ScalarMul
    vector of tests/golden/eip196_kat.json through the reference's own symbols, no GPU involved"""
    from porla_amd import multiexp as mx
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eip196_kat.json")))
    assert len(kat["add"]) >= 8 and len(kat["mul"]) >= 16
    for e in kat["add"]:
        assert mx.bn254_add(bytes.fromhex(e["a"]), bytes.fromhex(e["b"])).hex() == e["sum"], e["name"]
        assert mx.bn254_add(bytes.fromhex(e["b"]), bytes.fromhex(e["a"])).hex() == e["sum"], e["name"]
    for e in kat["mul"]:
        assert mx.bn254_mult(bytes.fromhex(e["p"]), bytes.fromhex(e["k"])).hex() == e["r"], e["name"]


def test_host_field_arithmetic_without_adx_instructions():
    """the x86-64 host pass multiplies with mulx / adcx / adox where the CPU has them (fe.hip.h:mont_mul4_adx,
    host_fold64.hpp:mul512_adx); PORLA_NO_ADX=1 takes the portable products instead -- the path of any other host.  The known
    answers, the tree fold and the pairing self-check must come out the same there (a child process: the switch is read once)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PORLA_NO_ADX="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(here, "test_abi_cpu.py"), os.path.join(here, "test_pairing_cpu.py"),
                        "-k", "eip196 or tree_fold or point_ops or kzg_host_side or bilinearity"],
                       env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_mult_point_against_the_c_oracle_on_random_and_edge_scalars():
    """mult_point runs signed 4-bit windows in 64-bit limbs on the host (host_fold64.hpp:h_scalar_mul64): every digit value, the
    carry into the 65th digit, scalars at and above the group order, the point at infinity -- against oracle/bn254_ref.c"""
    import random
    from porla_amd import multiexp as mx
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    pts = common.synth_points(6, start=4242)
    rnd = random.Random(20261004)
    scalars = [0, 1, 2, 7, 8, 9, 15, 16, 17, 0x88888888, R - 1, R, R + 1, (1 << 256) - 1, (1 << 255), int("8" * 64, 16), int("f" * 63 + "8", 16),
               int("7" * 64, 16), int("9" * 64, 16)] + [rnd.getrandbits(256) for _ in range(40)] + [rnd.getrandbits(64) for _ in range(10)]
    for i, k in enumerate(scalars):
        p = pts[64 * (i % 6):64 * (i % 6) + 64]
        kb = k.to_bytes(32, "big")
        assert mx.bn254_mult(p, kb) == common.oracle_msm(kb, p, 1), hex(k)
    assert mx.bn254_mult(bytes(64), (12345).to_bytes(32, "big")) == bytes(64)


def test_division_step_inversion_model_stays_inside_its_registers():
    """the finish kernels invert with 25 rounds of 30 Bernstein-Yang division steps on signed 30-bit limbs
    (inv30.hip.h:fe_inv_safegcd); tools/safegcd_model.py is that procedure on Python integers with every 32- / 64-bit
    register range asserted, against pow(V, -1, p) for both base fields -- edge values and random ones"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "safegcd_model.py"), "2000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "safegcd model: ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def test_unloadable_rccl_is_an_error_code_not_a_crash():
    """ADVICE r02 (dist.hip:load_rccl): a candidate that fails to dlopen must end in PORLA_ERR_STATE with the loader's message --
    dlerror() returns its message once and clears it, a second call appended NULL to a std::string.  PORLA_RCCL_LIB names THE
    library to bind, nothing else is tried.  A child process: the first successful bind is kept for the process's lifetime."""
    import subprocess
    import sys
    code = ("import ctypes, sys\n"
            "from porla_amd import lib\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc = lib.porla_dist_unique_id(buf)\n"
            "lib.porla_gpu_last_error.restype = ctypes.c_char_p\n"
            "print('rc', rc, lib.porla_gpu_last_error().decode())\n"
            "w = ctypes.c_int(-1)\n"
            "lib.porla_dist_info(None, ctypes.byref(w))\n"
            "print('world', w.value)\n")
    env = dict(os.environ, PORLA_RCCL_LIB="/nonexistent/librccl-not-here.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("rc ")][0]
    assert int(line.split()[1]) != 0                      # PORLA_ERR_STATE
    assert "librccl-not-here.so" in line and "RCCL not found" in line
    assert "world 0" in r.stdout


def test_sharded_commit_rows_takes_its_row_stride_from_the_caller_or_the_library():
    """ADVICE r02 (sharded.py): the row stride is 32 bytes x the SRS size, not a hard-coded 4096"""
    from porla_amd import sharded
    seen = []
    rows = bytes(range(256)) * 3                          # 6 rows of 4 coefficients (128 B each)
    lo, out = sharded.sharded_commit_rows(rows, 6, commit=lambda r, n: seen.append((bytes(r), n)) or b"x" * n)
    assert lo == 0 and out == b"x" * 6 and seen == [(rows, 6)]
    with pytest.raises(ValueError):
        sharded.sharded_commit_rows(rows, 6, commit=lambda r, n: b"", row_coefficients=128)
