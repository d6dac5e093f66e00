"""End-to-end KZG audit over the engine's kernels (GPU box only): the consistency the protocol itself checks.

Server side of porla/Server/Server.hpp:564-931 with every arithmetic step on the engine:
  blocks --(porla_icc_encode: CRebuild data part)--> encoded rows X            Server.hpp:1548-1687
  Commit(block_i) --(porla_icc_mac_encode: CRebuild MAC part)--> encoded MACs   Server.hpp:1590-1609
  challenge (idx, coeff) --(porla_audit_combine_device)--> B mod p_icc, alignment scalars c    :790-828, :531-541
  combined_MAC = compute_multi_exp(coeffs, MACs[idx])                          Server.hpp:900
  align_value  = compute_digest_from_srs(c)                                    Server.hpp:550-560
  proof        = create_proof(z, B)                                            Server.hpp:907 -> main.go:153-175
What must hold (Client::audit, porla/Client/Client.hpp:849-876, without the client's secret alpha / complements):
  proof.commitment == combined_MAC + align_value      and      verify_proof(proof) == 1
because the commitment is linear over Z_q and the code is linear: Commit(X_k mod q) = encoded MAC_k."""
import ctypes
import hashlib
import random

import pytest

from tests import common

pytestmark = pytest.mark.gpu

TAU = bytes.fromhex("ffeeddccbbaa99887766554433221100")     # TAU_KEY, config.hpp:39
ALPHA = bytes.fromhex("00112233445566778899aabbccddeeff")   # SECRET_KEY, config.hpp:38
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


@pytest.mark.parametrize("n_blocks,write_step,part", [(64, 0, 0), (256, 37, 1)])
def test_kzg_audit_is_consistent_end_to_end(n_blocks, write_step, part):
    import numpy as np
    import torch
    from porla_amd import icc, multiexp as mx
    ncols = 128
    mx.init_key(TAU, ALPHA)
    blob = mx.init_SRS(ncols)
    mx.init_SRS_from_data(ncols, blob)
    # blocks: chunk 0 = block id, the rest random 256-bit values (Client.hpp:367-372); 32-byte little-endian chunks
    rows = b""
    for i in range(n_blocks):
        rows += i.to_bytes(32, "little")
        rows += b"".join(hashlib.sha256(b"blk" + i.to_bytes(4, "little") + j.to_bytes(4, "little")).digest() for j in range(ncols - 1))
    # per-block commitments (the MAC without the client's alpha / complement): coefficients = chunks as big-endian scalars
    rows_be = b"".join(rows[32 * k:32 * k + 32][::-1] for k in range(n_blocks * ncols))
    macs_u = mx.kzg_commit_batch_host(rows_be, n_blocks)
    # encode data and MACs
    x_rows = icc.crebuild_host(rows, n_blocks, ncols, "bn254", write_step, part, want_aligned=False, want_scalars=False)[0]
    macs_h = icc.mac_crebuild_host(macs_u, n_blocks, "bn254", write_step, part)
    # spot check of the linearity the audit relies on: Commit(X_k mod q) == encoded MAC_k
    for k in (0, 1, n_blocks - 1):
        coeffs = b"".join((int.from_bytes(x_rows[64 * (k * ncols + j):64 * (k * ncols + j) + 64], "little") % R).to_bytes(32, "big")
                          for j in range(ncols))
        assert mx.compute_digest_from_srs(coeffs) == macs_h[64 * k:64 * k + 64]
    # challenge: NUM_CHECK_AUDIT-style random rows with abs(int32) coefficients (Server.hpp:604-621)
    rnd = random.Random(n_blocks)
    n_points = 128
    idx = [rnd.randrange(n_blocks) for _ in range(n_points)]
    coef = [rnd.getrandbits(31) for _ in range(n_points)]
    d_rows = torch.frombuffer(bytearray(x_rows), dtype=torch.uint8).cuda()
    d_idx = torch.tensor(idx, dtype=torch.int64).cuda()
    d_coef = torch.tensor(np.array(coef, dtype=np.uint32).view(np.int32)).cuda()
    d_b_be = torch.empty(32 * ncols, dtype=torch.uint8, device="cuda")
    d_c = torch.empty(32 * ncols, dtype=torch.uint8, device="cuda")
    icc.audit_combine_device(d_rows.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), n_points, 0, 0, 0, 0, ncols, "bn254",
                             d_aligned_be=d_b_be.data_ptr(), d_scalars=d_c.data_ptr(),
                             stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    b_be, c_be = bytes(d_b_be.cpu().numpy()), bytes(d_c.cpu().numpy())
    # the two MSMs of the audit (Server.hpp:900-901; the alignment MACs of a fresh level are infinity)
    scalars = b"".join(mx.bn254_scalar_set_int(v) for v in coef)
    points = b"".join(macs_h[64 * i:64 * i + 64] for i in idx)
    combined_mac = mx.bn254_multi_exp(points, scalars, n_points)
    align_value = mx.compute_digest_from_srs(c_be)              # align_MAC, Server.hpp:550-560
    z = 0x0123456789abcdef
    commitment, proof_h, point, claim = mx.create_proof(z, b_be)
    assert commitment == mx.bn254_add(combined_mac, align_value)
    assert mx.verify_proof(commitment, proof_h, point, claim)
    # the same audit as ONE call (porla_kzg_audit_device): every output equal to the step-by-step ones
    d_macs = torch.frombuffer(bytearray(macs_h), dtype=torch.uint8).cuda()
    d_align = torch.zeros(64 * n_blocks, dtype=torch.uint8, device="cuda")      # a fresh level: alignment MACs at infinity
    torch.cuda.synchronize()
    one = mx.kzg_audit_device(d_rows.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), n_points, 0, 0, 0, 0, d_macs.data_ptr(),
                              d_align.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), n_points, z)
    assert one["combined_mac"] == combined_mac and one["combined_align"] == bytes(64) and one["align_value"] == align_value
    assert (one["commitment"], one["proof_h"], one["point"], one["claim"]) == (commitment, proof_h, point, claim) and one["b"] == b_be
    # the stream contract (include/porla_gpu.h): the challenge uploaded ASYNCHRONOUSLY on the call's stream, queued behind a long
    # kernel, and NO host synchronisation before the call -- the pair's gather runs on a stream of the library's own and must still
    # see the uploaded indices (it raced with the upload before the audits ordered their streams behind the caller's)
    side = torch.cuda.Stream()
    h_idx = torch.tensor(idx, dtype=torch.int64).pin_memory()
    h_coef = torch.tensor(np.array(coef, dtype=np.uint32).view(np.int32)).pin_memory()
    d_idx2 = torch.zeros(n_points, dtype=torch.int64, device="cuda")            # zeros = a wrong challenge until the copy lands
    d_coef2 = torch.zeros(n_points, dtype=torch.int32, device="cuda")
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(4):
            big.normal_()                                                          # a few ms of work in front of the copies
        d_idx2.copy_(h_idx, non_blocking=True)
        d_coef2.copy_(h_coef, non_blocking=True)
        late = mx.kzg_audit_device(d_rows.data_ptr(), d_idx2.data_ptr(), d_coef2.data_ptr(), n_points, 0, 0, 0, 0, d_macs.data_ptr(),
                                   d_align.data_ptr(), d_idx2.data_ptr(), d_coef2.data_ptr(), n_points, z, stream=side.cuda_stream)
    assert late == one
    # a tampered row breaks it
    bad = bytearray(b_be)
    bad[31] ^= 1
    c2, h2, p2, y2 = mx.create_proof(z, bytes(bad))
    assert c2 != mx.bn254_add(combined_mac, align_value)


def test_one_call_audit_edges_and_two_threads():
    """porla_kzg_audit_device: an empty challenge (B = 0: every commitment is infinity, the claim 0), a challenge without MACs, and
    two threads calling at once (the call serialises itself: both get the single-threaded answer); the two-phase pair's error paths"""
    import threading
    import numpy as np
    import torch
    from porla_amd import icc, lib, multiexp as mx
    ncols, n_rows, n_points = 128, 64, 40
    mx.init_key(TAU, ALPHA)
    mx.init_SRS_from_data(ncols, mx.init_SRS(ncols))
    rnd = random.Random(77)
    lcm = R * (207 * 2 ** 248 + 1)
    store = b"".join(rnd.randrange(lcm).to_bytes(64, "little") for _ in range(n_rows * ncols))
    macs = common.synth_points(n_rows, start=500)
    d_rows = torch.frombuffer(bytearray(store), dtype=torch.uint8).cuda()
    d_macs = torch.frombuffer(bytearray(macs), dtype=torch.uint8).cuda()
    d_align = torch.frombuffer(bytearray(macs[64:] + macs[:64]), dtype=torch.uint8).cuda()
    idx = [rnd.randrange(n_rows) for _ in range(n_points)]
    coef = [rnd.getrandbits(31) for _ in range(n_points)]
    d_idx = torch.tensor(idx, dtype=torch.int64).cuda()
    d_coef = torch.tensor(np.array(coef, dtype=np.uint32).view(np.int32)).cuda()
    torch.cuda.synchronize()
    z = 99
    # empty challenge
    e = mx.kzg_audit_device(0, 0, 0, 0, 0, 0, 0, 0, d_macs.data_ptr(), d_align.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), 0, z)
    assert e["combined_mac"] == e["combined_align"] == e["align_value"] == e["commitment"] == e["proof_h"] == bytes(64)
    assert e["claim"] == bytes(32) and e["b"] == bytes(32 * ncols) and int.from_bytes(e["point"], "big") == z
    # the reference result of a real challenge, from the separate entry points
    d_b = torch.empty(32 * ncols, dtype=torch.uint8, device="cuda")
    d_c = torch.empty(32 * ncols, dtype=torch.uint8, device="cuda")
    icc.audit_combine_device(d_rows.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), n_points, 0, 0, 0, 0, ncols, "bn254",
                             d_aligned_be=d_b.data_ptr(), d_scalars=d_c.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    b_be, c_be = bytes(d_b.cpu().numpy()), bytes(d_c.cpu().numpy())
    sc = b"".join(c.to_bytes(32, "big") for c in coef)
    want = dict(combined_mac=common.oracle_msm(sc, b"".join(macs[64 * i:64 * i + 64] for i in idx), n_points),
                align_value=mx.compute_digest_from_srs(c_be), b=b_be)
    want["commitment"], want["proof_h"], want["point"], want["claim"] = mx.create_proof(z, b_be)
    args = (d_rows.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(), n_points, 0, 0, 0, 0, d_macs.data_ptr(), d_align.data_ptr(),
            d_idx.data_ptr(), d_coef.data_ptr(), n_points, z)
    got = [None, None]

    def worker(k):
        for _ in range(20):
            got[k] = mx.kzg_audit_device(*args)
    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for g in got:
        assert all(g[k] == v for k, v in want.items())
    # no MACs asked for: both sums are infinity, the rest unchanged
    nm = mx.kzg_audit_device(*args[:12], 0, z)
    assert nm["combined_mac"] == nm["combined_align"] == bytes(64) and nm["commitment"] == want["commitment"]
    # two-phase pair: argument and state errors
    vp = ctypes.c_void_p
    beg = lib.porla_bn254_audit_msm_pair_begin
    a = (vp(d_macs.data_ptr()), vp(d_align.data_ptr()), vp(d_idx.data_ptr()), vp(d_coef.data_ptr()))
    assert beg(0, *a, n_points, None) != 0 and beg(4, *a, n_points, None) != 0        # slots 1..3 only
    assert beg(1, *a, 0, None) != 0                                                    # nothing to do is an error here
    assert beg(1, *a, n_points, None) == 0
    assert beg(1, *a, n_points, None) != 0                                             # already begun
    assert mx.audit_msm_pair_end(1, "bn254")[0] == want["combined_mac"]


def test_ipa_audit_in_one_call():
    """porla_ipa_audit_device (Server::audit of the IPA build up to the inner-product proof, Server.hpp:790-857): B and the alignment
    scalars against the Python restatement, both MSMs and both Pedersen commitments over the generators against the oracle's secp256k1
    MSM; a 256-bit-row level mixed in"""
    import numpy as np
    import torch
    import icc_py
    from porla_amd import multiexp as mx
    ncols, n_rows, n64, n32 = 128, 96, 150, 50
    rnd = random.Random(857)
    lcm, p = icc_py.LCM["secp256k1"], icc_py.P_ICC
    s64 = [[rnd.randrange(lcm) for _ in range(ncols)] for _ in range(n_rows)]
    s32 = [[rnd.randrange(p) for _ in range(ncols)] for _ in range(n_rows)]
    gens = common.secp_bench_points(ncols)
    macs = common.secp_bench_points(2 * n_rows + 1)[64:]                 # some other valid points
    mac_store, align_store = macs[:64 * n_rows], macs[64 * n_rows:64 * 2 * n_rows]
    i64, i32 = [rnd.randrange(n_rows) for _ in range(n64)], [rnd.randrange(n_rows) for _ in range(n32)]
    c64, c32 = [rnd.getrandbits(31) for _ in range(n64)], [rnd.getrandbits(31) for _ in range(n32)]
    dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    d_s64 = dev(b"".join(v.to_bytes(64, "little") for r in s64 for v in r))
    d_s32 = dev(b"".join(v.to_bytes(32, "little") for r in s32 for v in r))
    d_i64, d_i32 = torch.tensor(i64, dtype=torch.int64).cuda(), torch.tensor(i32, dtype=torch.int64).cuda()
    d_c64 = torch.tensor(np.array(c64, dtype=np.uint32).view(np.int32)).cuda()
    d_c32 = torch.tensor(np.array(c32, dtype=np.uint32).view(np.int32)).cuda()
    # the MACs of the challenged rows: one index / coefficient array over both levels' rows (the server concatenates them)
    m_idx, m_coef = i64 + i32, c64 + c32
    d_mi = torch.tensor(m_idx, dtype=torch.int64).cuda()
    d_mc = torch.tensor(np.array(m_coef, dtype=np.uint32).view(np.int32)).cuda()
    d_ms, d_as = dev(mac_store), dev(align_store)
    torch.cuda.synchronize()
    fb = mx.FixedBase("secp256k1", gens, ncols)
    got = fb.ipa_audit_device(d_s64.data_ptr(), d_i64.data_ptr(), d_c64.data_ptr(), n64, d_s32.data_ptr(), d_i32.data_ptr(), d_c32.data_ptr(),
                              n32, ncols, d_ms.data_ptr(), d_as.data_ptr(), d_mi.data_ptr(), d_mc.data_ptr(), len(m_idx))
    B, mods, cs = icc_py.audit_combine([s64[i] for i in i64] + [s32[i] for i in i32], c64 + c32, "secp256k1")
    be = lambda vals: b"".join(v.to_bytes(32, "big") for v in vals)
    assert got["b"] == be(mods)
    sc = be(m_coef)
    assert got["combined_mac"] == common.oracle_secp_msm(sc, b"".join(mac_store[64 * i:64 * i + 64] for i in m_idx), len(m_idx))
    assert got["combined_align"] == common.oracle_secp_msm(sc, b"".join(align_store[64 * i:64 * i + 64] for i in m_idx), len(m_idx))
    assert got["align_value"] == common.oracle_secp_msm(be(cs), gens, ncols)
    assert got["commitment"] == common.oracle_secp_msm(be(mods), gens, ncols)
    # wrong kind of fixed base
    bn = mx.FixedBase("bn254", common.synth_points(4), 4)
    with pytest.raises(RuntimeError):
        bn.ipa_audit_device(0, 0, 0, 0, 0, 0, 0, 0, 4, d_ms.data_ptr(), d_as.data_ptr(), d_mi.data_ptr(), d_mc.data_ptr(), 1)
