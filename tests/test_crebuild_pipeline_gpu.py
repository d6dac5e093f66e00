"""GPU box: the last encode stage of a CRebuild as ONE device-resident chain (tools/bench_crebuild.py; bench.py --workload crebuild)
-- porla_icc_encode_device -> porla_kzg_commit_batch_device (align_MAC, porla/Server/Server.hpp:531-560) +
porla_icc_mac_encode_device, X and Y parts (Server.hpp:1487-1833, :2059-2065), one stream, no host synchronisation inside --
against the oracle chain, and the protocol's own consistency between its three outputs."""
import ctypes
import json
import os
import subprocess
import sys

import pytest

from tests import common

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(common.ROOT, "tools"))
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
P_ICC = 207 * 2**248 + 1


@pytest.mark.parametrize("log2rows,write_step", [(6, 0), (8, 37), (11, 1000003)])
def test_chain_matches_the_oracle_chain(log2rows, write_step):
    import torch
    import bench_crebuild as bc
    from porla_amd import icc, multiexp as mx
    n = 1 << log2rows
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    mx.init_key(bc.TAU, bc.ALPHA)
    mx.init_SRS_from_data(bc.NCOLS, mx.init_SRS(bc.NCOLS))
    g = torch.Generator(device=dev).manual_seed(7 + log2rows)
    d_rows = torch.randint(0, 256, (n * bc.NCOLS * 32,), dtype=torch.uint8, device=dev, generator=g)
    d_macs = torch.empty(64 * n, dtype=torch.uint8, device=dev)
    mx.kzg_commit_batch_device(d_rows.data_ptr(), n, d_macs.data_ptr(), stream)
    bufs = bc.alloc(torch, n, dev)
    side = torch.cuda.Stream(device=dev)
    if log2rows == 8:       # the calls in the reference's order, one MAC encode per part, one stream
        bc.run_chain(torch, icc, mx, d_rows, d_macs, n, write_step, bufs, stream, reference_order=True)
    elif log2rows == 11:    # the whole stage as ONE call (porla_kzg_crebuild_stage_device: MAC network first, commitments in the
        #                     two-waves-per-SIMD form of their kernel beside it)
        icc.kzg_crebuild_stage_device(d_rows.data_ptr(), n, write_step, bufs[0][0].data_ptr(), bufs[1][0].data_ptr(), bufs[0][1].data_ptr(),
                                      bufs[0][2].data_ptr(), d_macs.data_ptr(), bufs[0][3].data_ptr(), bufs[1][3].data_ptr(), stream)
    else:                   # both MAC halves from one network, on a second stream beside the data side
        bc.run_chain(torch, icc, mx, d_rows, d_macs, n, write_step, bufs, stream, side.cuda_stream)
    torch.cuda.synchronize()
    rows, macs = bytes(d_rows.cpu().numpy()), bytes(d_macs.cpu().numpy())
    L = common.oracle()
    L.oracle_kzg_init_key(bc.TAU, ctypes.c_size_t(16), bc.ALPHA, ctypes.c_size_t(16))
    L.oracle_kzg_init_srs(ctypes.c_size_t(bc.NCOLS), (1).to_bytes(32, "big"))
    srs = ctypes.create_string_buffer(64 * bc.NCOLS)
    L.oracle_kzg_srs_g1_raw(srs)
    for part in (0, 1):
        x = ctypes.create_string_buffer(64 * n * bc.NCOLS)
        al = ctypes.create_string_buffer(32 * n * bc.NCOLS)
        sc = ctypes.create_string_buffer(32 * n * bc.NCOLS)
        L.oracle_icc_crebuild(rows, ctypes.c_size_t(n), ctypes.c_size_t(bc.NCOLS), 0, part, ctypes.c_uint64(write_step), x, al, sc,
                              common.ncpu())
        mh = ctypes.create_string_buffer(64 * n)
        L.oracle_icc_mac_crebuild(macs, ctypes.c_size_t(n), 0, part, ctypes.c_uint64(write_step), mh, common.ncpu())
        got = [bytes(t.cpu().numpy()) for t in bufs[part]]
        assert got[0] == al.raw, "aligned rows, part %d" % part
        assert got[1] == sc.raw, "alignment scalars, part %d" % part
        assert got[2] == common.oracle_commit_batch("bn254", sc.raw, n, bc.NCOLS, srs.raw), "align_MAC commitments, part %d" % part
        assert got[3] == mh.raw, "encoded MACs, part %d" % part
        # the identity align_MAC exists for (Server.hpp:531-560): with A the encoded value mod LCM, aligned = A mod p_icc and
        # c = (aligned - A) mod q, so aligned = A + c (mod q) and, the commitment being linear over Z_q,
        #     Commit(aligned row mod q) = Commit(A row mod q) + Commit(c row)
        # row 1 and the last row: the left side through compute_digest_from_srs, the right side from the oracle's A and the chain's
        # align_MAC output
        for k in (1, n - 1):
            a_row = [int.from_bytes(x.raw[64 * (k * bc.NCOLS + j):64 * (k * bc.NCOLS + j) + 64], "little") for j in range(bc.NCOLS)]
            al_row = [int.from_bytes(got[0][32 * (k * bc.NCOLS + j):32 * (k * bc.NCOLS + j) + 32], "little") for j in range(bc.NCOLS)]
            assert al_row == [a % P_ICC for a in a_row]
            lhs = mx.compute_digest_from_srs(b"".join((v % R).to_bytes(32, "big") for v in al_row))
            rhs = mx.bn254_add(mx.compute_digest_from_srs(b"".join((a % R).to_bytes(32, "big") for a in a_row)), got[2][64 * k:64 * k + 64])
            assert lhs == rhs


def test_bench_line_of_the_crebuild_workload():
    """bench.py --workload crebuild: one JSON line in the bench contract, the sample chain bit-exact, the per-kernel split present"""
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--workload", "crebuild", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=common.ROOT,
                       env=dict(os.environ, PORLA_CREBUILD_LOG2ROWS="12"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["unit"] == "rows/s" and d["value"] > 0 and d["bit_exact_vs_oracle"] is True
    assert d["config"]["rows"] == 1 << 12 and d["config"]["commitments_per_row"] == 2
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    ks = d["kernels_ms_per_step"]
    assert any("icc" in k for k in ks) and any("fb_commit" in k for k in ks) and any("mac" in k for k in ks)
    assert d["sum_kernels_ms_per_step"] > 0 and d["roofline"]["frac"] > 0
    assert "porla_kzg_crebuild_stage_device" in d["entry_point"] and d["separate_calls_two_streams_ms_per_step"] > 0
