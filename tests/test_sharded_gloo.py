"""CPU, world_size 2 over gloo: the N>1 path of the MSM (all_gather of the 96-byte partial Jacobians + fold) is
exercised with per-rank partials produced by the oracle standing in for the device MSM (no GPU here); the exchange
and fold code is the one bench.py runs over RCCL."""
import os
import subprocess
import sys

from tests import common

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["PORLA_ROOT"])
import torch.distributed as dist
from tests import common
from porla_amd import sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 600
sc, pt = common.synth_inputs(n)
lo, hi = n * rank // world, n * (rank + 1) // world
if os.environ.get("PORLA_EMPTY_RANK") == str(rank):
    part_aff = bytes(64)   # a rank whose shard sums to infinity
    lo = hi
else:
    part_aff = common.oracle_msm(sc[32 * lo:32 * hi], pt[64 * lo:64 * hi], hi - lo, threads=1)
parts = sharded.gather_partials(sharded.affine_to_partial(part_aff))
assert len(parts) == world
got = sharded.fold_partials("bn254", parts)
if os.environ.get("PORLA_EMPTY_RANK") is None:
    want = common.oracle_msm(sc, pt, n, threads=1)
else:
    e = int(os.environ["PORLA_EMPTY_RANK"])
    keep = [r for r in range(world) if r != e]
    import bn254_py as o
    acc = bytes(64)
    for r in keep:
        l2, h2 = n * r // world, n * (r + 1) // world
        acc = o.add_point(acc, common.oracle_msm(sc[32 * l2:32 * h2], pt[64 * l2:64 * h2], h2 - l2, threads=1))
    want = acc
assert got == want, (got.hex(), want.hex())
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def run(world, extra_env=None):
    env = dict(os.environ)
    env.update({"PORLA_ROOT": common.ROOT, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29571", "WORLD_SIZE": str(world),
                "PORLA_NO_TORCH": "0", "OMP_NUM_THREADS": "1"})
    env.update(extra_env or {})
    procs = []
    for r in range(world):
        e = dict(env)
        e["RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return outs


def test_two_ranks_gather_and_fold():
    common.oracle()  # build once before forking
    outs = run(2)
    assert all("ok" in o for o in outs)


def test_two_ranks_one_shard_is_infinity():
    outs = run(2, {"PORLA_EMPTY_RANK": "1", "MASTER_PORT": "29572"})
    assert all("ok" in o for o in outs)


# ---- the paths that shard with no collective: commitment rows and ICC columns, world_size 2 over gloo
RANGE_WORKER = r"""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.environ["PORLA_ROOT"])
import torch.distributed as dist
from tests import common
from porla_amd import sharded, multiexp as mx
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
L = common.oracle()

# (1) batched commitments: rows are independent; rank g commits rows [g R / G, (g+1) R / G) -- the oracle stands in for the device
n_rows, n_coeffs = 7, 16
base = common.synth_points(n_coeffs)
rows = b"".join(hashlib.sha256(b"row" + i.to_bytes(4, "little")).digest() for i in range(n_rows * n_coeffs))
commit = lambda r, n: common.oracle_commit_batch("bn254", r, n, n_coeffs, base, threads=1)
lo, hi = sharded.my_range(n_rows)
mine = commit(rows[32 * n_coeffs * lo:32 * n_coeffs * hi], hi - lo)
parts = sharded.gather_objects((lo, mine))            # bookkeeping for the check only: the data path exchanges nothing
assert [p[0] for p in parts] == [mx.shard_range(n_rows, r, world)[0] for r in range(world)]
assert b"".join(p[1] for p in parts) == commit(rows, n_rows)

# (2) ICC encode: the columns are independent transforms; rank g takes columns [g C / G, (g+1) C / G) of the row-major rows
n, ncols = 16, 6
raw = b"".join(hashlib.sha256(b"icc" + i.to_bytes(4, "little")).digest() for i in range(n * ncols))
def encode(rows_bytes, nc):
    x, al, sc = (ctypes.create_string_buffer(64 * n * nc), ctypes.create_string_buffer(32 * n * nc), ctypes.create_string_buffer(32 * n * nc))
    L.oracle_icc_crebuild(rows_bytes, ctypes.c_size_t(n), ctypes.c_size_t(nc), 0, 0, ctypes.c_uint64(0), x, al, sc, 1)
    return x.raw, al.raw, sc.raw
c0, c1 = sharded.my_range(ncols)
cols = b"".join(raw[32 * (r * ncols + c0):32 * (r * ncols + c1)] for r in range(n))      # the strided gather of the upload
x, al, sc = encode(cols, c1 - c0)
fx, fal, fsc = encode(raw, ncols)
for r in range(n):
    assert x[64 * r * (c1 - c0):64 * (r + 1) * (c1 - c0)] == fx[64 * (r * ncols + c0):64 * (r * ncols + c1)]
    assert al[32 * r * (c1 - c0):32 * (r + 1) * (c1 - c0)] == fal[32 * (r * ncols + c0):32 * (r * ncols + c1)]
    assert sc[32 * r * (c1 - c0):32 * (r + 1) * (c1 - c0)] == fsc[32 * (r * ncols + c0):32 * (r * ncols + c1)]
cover = sharded.gather_objects((c0, c1))
assert cover[0][0] == 0 and cover[-1][1] == ncols and all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_ranks_row_and_column_ranges_need_no_collective():
    common.oracle()
    env = dict(os.environ)
    env.update({"PORLA_ROOT": common.ROOT, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29573", "WORLD_SIZE": "2", "OMP_NUM_THREADS": "1"})
    procs = [subprocess.Popen([sys.executable, "-c", RANGE_WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o


def test_shard_range_rule():
    from porla_amd import multiexp as mx
    for n in (0, 1, 7, 128, 1 << 24):
        for world in (1, 2, 3, 8):
            ranges = [mx.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in ranges) - min(e - b for b, e in ranges) <= 1
    assert [mx.shard_range(128, g, 8) for g in range(8)] == [(16 * g, 16 * g + 16) for g in range(8)]     # 16 columns per GPU
