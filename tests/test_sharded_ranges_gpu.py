"""GPU box: the range splits that need no collective, through the C ABI -- ICC column ranges
(porla_icc_encode_cols_host / _host_multi; the reference splits the columns of a stage over 8 pool threads,
Server.hpp:1564-1686) and commitment row ranges (porla_kzg_commit_batch_host_multi; Server.hpp:1077-1078, 2061-2062) -- and the
same with two PROCESSES sharing this box's GPU, each on its own range."""
import ctypes
import hashlib
import os
import socket
import subprocess
import sys

import pytest

from tests import common

pytestmark = pytest.mark.gpu
TAU = bytes.fromhex("ffeeddccbbaa99887766554433221100")
ALPHA = bytes.fromhex("00112233445566778899aabbccddeeff")


def sha(seed, n):
    return b"".join(hashlib.sha256(seed + i.to_bytes(4, "little")).digest() for i in range(n))


def oracle_icc(rows, n, ncols, curve=0, part=0, write_step=0):
    L = common.oracle()
    x, al, sc = (ctypes.create_string_buffer(64 * n * ncols), ctypes.create_string_buffer(32 * n * ncols), ctypes.create_string_buffer(32 * n * ncols))
    L.oracle_icc_crebuild(rows, ctypes.c_size_t(n), ctypes.c_size_t(ncols), curve, part, ctypes.c_uint64(write_step), x, al, sc, common.ncpu())
    return x.raw, al.raw, sc.raw


@pytest.mark.parametrize("curve", ["bn254", "secp256k1"])
def test_icc_column_ranges_assemble_the_full_encode(curve):
    from porla_amd import icc, multiexp as mx
    n, ncols = 256, 128
    rows = sha(b"cols", n * ncols)
    want = oracle_icc(rows, n, ncols, icc.CURVE[curve])
    for G in (1, 3, 8):
        bufs = [ctypes.create_string_buffer(64 * n * ncols), ctypes.create_string_buffer(32 * n * ncols), ctypes.create_string_buffer(32 * n * ncols)]
        for g in range(G):                                    # what G ranks (or G devices) would each do
            c0, c1 = mx.shard_range(ncols, g, G)
            icc.crebuild_cols_host(rows, n, ncols, c0, c1, bufs[0], bufs[1], bufs[2], curve=curve)
        assert (bufs[0].raw, bufs[1].raw, bufs[2].raw) == want
    # a range writes ONLY its own columns
    x = ctypes.create_string_buffer(b"\xaa" * (64 * n * ncols), 64 * n * ncols)
    icc.crebuild_cols_host(rows, n, ncols, 16, 32, x, None, None, curve=curve)
    for r in (0, 100, 255):
        row = x.raw[64 * r * ncols:64 * (r + 1) * ncols]
        assert row[:64 * 16] == b"\xaa" * (64 * 16) and row[64 * 32:] == b"\xaa" * (64 * 96)
        assert row[64 * 16:64 * 32] == want[0][64 * (r * ncols + 16):64 * (r * ncols + 32)]
    # every visible device of this process, one host thread each
    assert icc.crebuild_host_multi(rows, n, ncols, curve=curve, devices=0) == want
    # Y part with a write step, through the column form
    wy = oracle_icc(rows, n, ncols, icc.CURVE[curve], 1, 77)
    y = ctypes.create_string_buffer(64 * n * ncols)
    for g in range(4):
        c0, c1 = mx.shard_range(ncols, g, 4)
        icc.crebuild_cols_host(rows, n, ncols, c0, c1, y, None, None, curve=curve, write_step=77, part=1)
    assert y.raw == wy[0]


def kzg_setup(mx):
    mx.init_key(TAU, ALPHA)
    mx.init_SRS_from_data(128, mx.init_SRS(128))
    o = common.oracle()
    o.oracle_kzg_init_key(TAU, ctypes.c_size_t(16), ALPHA, ctypes.c_size_t(16))
    o.oracle_kzg_init_srs(ctypes.c_size_t(128), (1).to_bytes(32, "big"))
    raw = ctypes.create_string_buffer(64 * 128)
    o.oracle_kzg_srs_g1_raw(raw)
    return raw.raw


def test_commit_row_ranges_over_the_devices_of_this_process():
    from porla_amd import multiexp as mx, lib
    srs = kzg_setup(mx)
    n_rows = 1000
    rows = sha(b"rowrange", n_rows * 128)
    want = common.oracle_commit_batch("bn254", rows, n_rows, 128, srs)
    G = lib.porla_gpu_device_count()
    assert mx.kzg_commit_batch_host_multi(rows, n_rows, devices=G) == want
    assert mx.kzg_commit_batch_host_multi(rows, n_rows, devices=0) == want
    assert mx.kzg_commit_batch_host_multi(rows[:4096 * 3], 3, devices=0) == want[:64 * 3]        # fewer rows than devices is fine
    assert mx.kzg_commit_batch_host_multi(b"", 0, devices=0) == b""


RANK_WORKER = r"""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.environ["PORLA_ROOT"])
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank % torch.cuda.device_count())
dist.init_process_group("gloo", rank=rank, world_size=world)
from porla_amd import multiexp as mx, icc, sharded
from tests import common
sha = lambda seed, n: b"".join(hashlib.sha256(seed + i.to_bytes(4, "little")).digest() for i in range(n))
mx.init_key(bytes.fromhex("ffeeddccbbaa99887766554433221100"), bytes.fromhex("00112233445566778899aabbccddeeff"))
blob = [mx.init_SRS(128) if rank == 0 else None]
dist.broadcast_object_list(blob, src=0)                      # the server receives the SRS from the client (Server.hpp:183-188)
mx.init_SRS_from_data(128, blob[0])
n_rows = 301
rows = sha(b"two-rank-rows", n_rows * 128)
lo, mine = sharded.sharded_commit_rows(rows, n_rows)         # this rank's rows on the engine
parts = sharded.gather_objects((lo, mine))
if rank == 0:
    o = common.oracle()
    o.oracle_kzg_init_key(bytes.fromhex("ffeeddccbbaa99887766554433221100"), ctypes.c_size_t(16), bytes.fromhex("00112233445566778899aabbccddeeff"), ctypes.c_size_t(16))
    o.oracle_kzg_init_srs(ctypes.c_size_t(128), (1).to_bytes(32, "big"))
    raw = ctypes.create_string_buffer(64 * 128); o.oracle_kzg_srs_g1_raw(raw)
    assert b"".join(p[1] for p in parts) == common.oracle_commit_batch("bn254", rows, n_rows, 128, raw.raw)
n, ncols = 64, 128
data = sha(b"two-rank-icc", n * ncols)
c0, c1 = sharded.my_range(ncols)
x = ctypes.create_string_buffer(64 * n * ncols)
icc.crebuild_cols_host(data, n, ncols, c0, c1, x, None, None)
cols = sharded.gather_objects((c0, c1, b"".join(x.raw[64 * (r * ncols + c0):64 * (r * ncols + c1)] for r in range(n))))
if rank == 0:
    L = common.oracle()
    fx = ctypes.create_string_buffer(64 * n * ncols)
    L.oracle_icc_crebuild(data, ctypes.c_size_t(n), ctypes.c_size_t(ncols), 0, 0, ctypes.c_uint64(0), fx, None, None, 4)
    for a, b, blk in cols:
        for r in range(n):
            assert blk[64 * r * (b - a):64 * (r + 1) * (b - a)] == fx.raw[64 * (r * ncols + a):64 * (r * ncols + b)]
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_processes_each_on_its_own_range():
    common.oracle()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PORLA_ROOT=common.ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PORLA_COMMIT_TABLE_GB="2")
    procs = [subprocess.Popen([sys.executable, "-c", RANK_WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o
