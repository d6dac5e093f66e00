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
