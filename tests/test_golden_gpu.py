"""GPU box: the committed fixtures fed STRAIGHT through the HIP entry points -- no oracle in between.

tests/golden/bn254_golden.json (MSM cases n = 1 .. 3 200, the KZG digests and openings; produced by the Python big-int
restatement of porla/main.go:70-175, script alongside) and tests/golden/icc_golden.json (CRebuild_Cached X / Y parts, rows mod
p_icc, alignment scalars; porla/Server/Server.hpp:1487-1833, :531-541) pin the oracles on the CPU (tests/test_oracle_bn254.py,
test_oracle_icc.py); here the same bytes go through compute_multi_exp, the device-pointer MSM, compute_digest_from_srs /
create_proof / the commitment batches, and porla_icc_encode_*, and must come back as the fixture says.  (The secp256k1 and MAC
fixtures already go through the engine in test_msm_secp256k1_gpu.py / test_mac_fft_gpu.py.)"""
import json
import os

import pytest

from tests import common

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(__file__)
BN = json.load(open(os.path.join(HERE, "golden", "bn254_golden.json")))
ICC = json.load(open(os.path.join(HERE, "golden", "icc_golden.json")))


@pytest.fixture(params=["default", "plain", "general"])
def msm_path(request):
    """default: the engine's own choice (single-launch path at these sizes, GLV split where it pays); plain: no scalar split;
    general: the multi-kernel path at every size"""
    from porla_amd import lib
    lib.porla_gpu_set_msm_glv(0 if request.param == "plain" else -1)
    lib.porla_gpu_set_msm_small(0 if request.param == "general" else 1, 0)
    yield request.param
    lib.porla_gpu_set_msm_glv(-1)
    lib.porla_gpu_set_msm_small(1, 0)


@pytest.mark.parametrize("case", BN["msm"], ids=lambda c: c["name"])
def test_bn254_msm_fixture_through_every_entry_point(case, msm_path):
    import torch
    from porla_amd import multiexp as mx
    sc, pt, n = bytes.fromhex(case["scalars"]), bytes.fromhex(case["points"]), case["n"]
    assert mx.bn254_multi_exp(pt, sc, n).hex() == case["result"]                 # compute_multi_exp (main.go:118-138)
    assert mx.msm_host("bn254", sc, pt, n).hex() == case["result"]               # porla_bn254_msm_host
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    assert mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, stream).hex() == case["result"]
    mx.msm_begin(1, d_sc.data_ptr(), d_pt.data_ptr(), n, stream)                 # the two-phase form
    assert mx.msm_end(1).hex() == case["result"]
    if n >= 2:                                                                    # two partial Jacobians, folded
        h = n // 2
        parts = (mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), h, stream, partial=True)
                 + mx.msm_device("bn254", d_sc.data_ptr() + 32 * h, d_pt.data_ptr() + 64 * h, n - h, stream, partial=True))
        assert mx.jac_sum("bn254", parts, 2).hex() == case["result"]


def test_kzg_fixture_through_the_plugin_symbols_and_the_batches():
    import torch
    from porla_amd import multiexp as mx
    kz = BN["kzg"]
    tau, alpha, n = bytes.fromhex(kz["tau"]), bytes.fromhex(kz["alpha"]), kz["n"]
    mx.init_key(tau, alpha)
    blob = mx.init_SRS(n)                                                         # client side
    assert blob[:4 + 32 * n].hex() == kz["srs_g1_blob"]
    assert blob.hex() == kz["srs_blob"]       # all 32 n + 132 bytes (main.go:42-50, Client.hpp:350-357): the G2 half included
    for side in ("client", "server"):
        if side == "server":
            mx.init_SRS_from_data(n, blob)                                        # Server.hpp:183-188
        rows = b"".join(bytes.fromhex(c["f"]) for c in kz["cases"])
        want = b"".join(bytes.fromhex(c["digest_from_srs"]) for c in kz["cases"])
        for c in kz["cases"]:
            f = bytes.fromhex(c["f"])
            assert mx.compute_digest_from_srs(f).hex() == c["digest_from_srs"]     # main.go:103-116, k_fb_commit_small
            for op in c["open"]:
                got = mx.create_proof(op["z"], f)                                 # main.go:153-175: both commitments on the GPU
                assert [g.hex() for g in got] == [op["commitment"], op["H"], op["point"], op["claim"]]
                assert mx.verify_proof(*got)
        assert mx.kzg_commit_batch_host(rows, len(kz["cases"])) == want
        reps = 80                                                                 # > 64 rows: the batch kernels (k_fb_commit + fold + finish)
        assert mx.kzg_commit_batch_host(rows * reps, reps * len(kz["cases"])) == want * reps
        d_rows = torch.frombuffer(bytearray(rows * reps), dtype=torch.uint8).cuda()
        d_out = torch.empty(64 * reps * len(kz["cases"]), dtype=torch.uint8, device="cuda")
        mx.kzg_commit_batch_device(d_rows.data_ptr(), reps * len(kz["cases"]), d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert bytes(d_out.cpu().numpy()) == want * reps
    # client side again (the digests need alpha): compute_digest per row on the host, and the device batch of Client::initialize
    mx.init_SRS(n)
    rows = b"".join(bytes.fromhex(c["f"]) for c in kz["cases"])
    d_rows = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    d_out = torch.empty(64 * len(kz["cases"]), dtype=torch.uint8, device="cuda")
    mx.kzg_digest_batch_device(d_rows.data_ptr(), len(kz["cases"]), d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bytes(d_out.cpu().numpy()).hex() == "".join(c["digest"] for c in kz["cases"])
    for c in kz["cases"]:
        assert mx.compute_digest(bytes.fromhex(c["f"])).hex() == c["digest"]


@pytest.mark.parametrize("case", ICC["cases"], ids=lambda c: "%s-%d" % (c["curve"], c["n"]))
def test_icc_fixture_through_the_encode_entry_points(case):
    import torch
    from porla_amd import icc
    n, ncols, curve, ws = case["n"], case["ncols"], case["curve"], case["write_step"]
    rows = b"".join(bytes.fromhex(v) for r in case["rows"] for v in r)
    flat = lambda key: b"".join(bytes.fromhex(v) for r in case[key] for v in r)
    x, al, sc = icc.crebuild_host(rows, n, ncols, curve, ws, 0, scalar_le=True)
    assert x == flat("X") and al == flat("X_mod_p_icc") and sc == flat("X_align_scalars")
    # big-endian scalars (the bn254_scalar / secp256k1 wire form) are the same numbers
    sc_be = icc.crebuild_host(rows, n, ncols, curve, ws, 0, want_x=False, want_aligned=False)[2]
    assert b"".join(sc_be[32 * i:32 * i + 32][::-1] for i in range(n * ncols)) == flat("X_align_scalars")
    assert icc.crebuild_host(rows, n, ncols, curve, ws, 1, want_aligned=False, want_scalars=False)[0] == flat("Y")
    # the device-pointer form, outputs one at a time (each subset of outputs is its own path through the finish step)
    d_in = torch.frombuffer(bytearray(rows), dtype=torch.uint8).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    for key, width, kw in (("X", 64, "d_x"), ("X_mod_p_icc", 32, "d_aligned"), ("X_align_scalars", 32, "d_scalars")):
        d_out = torch.empty(width * n * ncols, dtype=torch.uint8, device="cuda")
        icc.crebuild_device(d_in.data_ptr(), n, ncols, curve, ws, 0, scalar_le=True, stream=stream, **{kw: d_out.data_ptr()})
        torch.cuda.synchronize()
        assert bytes(d_out.cpu().numpy()) == flat(key), key
    # column ranges (the multi-GPU split, Server.hpp:1564-1686) reassemble the fixture
    import ctypes
    bx, ba, bs = (ctypes.create_string_buffer(64 * n * ncols), ctypes.create_string_buffer(32 * n * ncols),
                  ctypes.create_string_buffer(32 * n * ncols))
    for c0, c1 in ((0, 1), (1, 3), (3, ncols)):
        icc.crebuild_cols_host(rows, n, ncols, c0, c1, bx, ba, bs, curve, ws, 0, scalar_le=True)
    assert bx.raw == flat("X") and ba.raw == flat("X_mod_p_icc") and bs.raw == flat("X_align_scalars")
