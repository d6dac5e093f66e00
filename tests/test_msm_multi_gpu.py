"""GPU box: the range-sharded MSM BEHIND THE C ABI (porla_*_msm_host_multi, and compute_multi_exp above its split threshold) --
BASELINE config 3's product path: host buffers cut into pair ranges, one host thread / stream / workspace per device, range
totals folded on the host like the reference folds its 8 pool threads' partial sums (porla/Client/Client.hpp:761-787).
Bar: the same 64 bytes as the oracle for every (shards, devices), at 2^22 and at the full 2^24 of config 3.
Also the multi-process form: the 96-byte partials through ncclAllGather issued from C++ (porla_dist_*)."""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mx():
    from porla_amd import multiexp
    return multiexp


@pytest.fixture(scope="module")
def inputs():
    return common.synth_inputs(1 << 14)


def device_count():
    from porla_amd import lib
    return lib.porla_gpu_device_count()


def big_inputs(log2n, seed):
    """2^log2n pairs: uniformly random 256-bit scalars (81 % of them >= r), points = 2^14 distinct ones tiled"""
    n, distinct = 1 << log2n, 1 << 14
    pt = common.synth_points(distinct) * (n // distinct)
    sc = np.random.default_rng(seed).bytes(32 * n)
    return sc, pt, n


@pytest.mark.parametrize("shards", [1, 2, 3, 4, 7, 9])
@pytest.mark.parametrize("n", [1, 5, 5000])
def test_forced_shards_on_one_device(mx, inputs, shards, n):
    sc, pt = inputs
    got = mx.msm_host_multi("bn254", sc[:32 * n], pt[:64 * n], n, shards=shards, devices=1)
    assert got == common.oracle_msm(sc, pt, n)
    s, d = mx.last_msm_multi()
    assert d == 1 and s == min(shards, n)


def test_empty_and_degenerate(mx, inputs):
    sc, pt = inputs
    assert mx.msm_host_multi("bn254", b"", b"", 0, shards=4, devices=1) == bytes(64)
    # every range sums to infinity (zero scalars), and a range of infinity points between ordinary ones
    n = 4096
    assert mx.msm_host_multi("bn254", bytes(32 * n), pt[:64 * n], n, shards=4, devices=1) == bytes(64)
    pts = pt[:64 * 1024] + bytes(64 * 1024) + pt[64 * 2048:64 * n]
    assert mx.msm_host_multi("bn254", sc[:32 * n], pts, n, shards=4, devices=1) == common.oracle_msm(sc, pts, n)


def test_secp256k1_forced_shards(mx):
    n = 3000
    sc, pt = common.secp_bench_scalars(n), common.secp_bench_points(n)
    want = common.oracle_secp_msm(sc, pt, n)
    for shards in (1, 4, 5):
        assert mx.msm_host_multi("secp256k1", sc, pt, n, shards=shards, devices=1) == want


def test_every_visible_device(mx, inputs):
    """G = device_count (1 on the builder's boxes, 8 on the driver's node); explicit and automatic shard counts"""
    sc, pt = inputs
    n, G = 1 << 14, device_count()
    want = common.oracle_msm(sc, pt, n)
    assert mx.msm_host_multi("bn254", sc, pt, n, shards=G, devices=G) == want
    assert mx.last_msm_multi() == (G, G)
    assert mx.msm_host_multi("bn254", sc, pt, n, shards=2 * G + 1, devices=G) == want
    assert mx.msm_host_multi("bn254", sc, pt, n, shards=0, devices=0) == want


def test_2p22_four_shards_one_device_and_all_devices(mx):
    sc, pt, n = big_inputs(22, 22)
    want = common.oracle_msm(sc, pt, n)
    assert mx.msm_host_multi("bn254", sc, pt, n, shards=4, devices=1) == want
    G = device_count()
    assert mx.msm_host_multi("bn254", sc, pt, n, shards=0, devices=G) == want
    assert mx.last_msm_multi()[1] == G
    # the reference's own boundary takes the same path from 2^18 pairs on
    assert mx.bn254_multi_exp(pt, sc, n) == want
    assert mx.last_msm_multi()[0] >= 2


def test_2p24_config3_whole_job(mx):
    """BASELINE config 3: one 2^24-pair MSM, range-sharded over every visible device (and over 4 ranges per device)"""
    sc, pt, n = big_inputs(24, 24)
    want = common.oracle_msm(sc, pt, n)
    G = device_count()
    assert mx.msm_host_multi("bn254", sc, pt, n, shards=4 * G, devices=G) == want
    assert mx.last_msm_multi() == (4 * G, G)
    assert mx.bn254_multi_exp(pt, sc, n) == want     # compute_multi_exp, automatic split
    if G == 1:
        # the same job device-resident on one GPU: ranges of 2^22 pairs into one bucket array (17-bit windows, the sort's staged
        # path), and as ONE pass with 20-bit windows (the sort's scattered-store fallback) -- msm_impl.hip.h:msm_launch
        import time
        import torch
        d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
        d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
        s = torch.cuda.current_stream().cuda_stream
        assert mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s) == want
        assert mx.last_msm_shape()[0] == 17
        t0 = time.perf_counter()
        mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
        t_ranges = time.perf_counter() - t0
        from porla_amd import lib
        lib.porla_gpu_set_msm_window(20)
        try:
            assert mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s) == want
            assert mx.last_msm_shape()[0] == 20
            t0 = time.perf_counter()
            mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s)
            t_single = time.perf_counter() - t0
        finally:
            lib.porla_gpu_set_msm_window(0)
        print("2^24 pairs, device-resident: %.2f ms in 4 ranges (c = 17), %.2f ms in one pass (c = 20)" % (t_ranges * 1e3, t_single * 1e3))
        lib.porla_gpu_release_msm_workspaces()


def test_device_resident_ranges_share_one_bucket_array(mx):
    """above 2^22 pairs a device-resident MSM runs as ranges into one bucket array (msm_impl.hip.h:msm_launch): 2^22 + 12 345 pairs =
    two ranges; the same scalar repeated 300 000 times across the range boundary makes one bucket per window a multi-item
    bucket in BOTH ranges (k_bucket_combine adds the earlier range's sum), and a block of zero scalars leaves buckets that
    only one range touches"""
    import torch
    n = (1 << 22) + 12345
    distinct = 1 << 14
    pt = (common.synth_points(distinct) * (n // distinct + 1))[:64 * n]
    sc = bytearray(np.random.default_rng(7).bytes(32 * n))
    lo = (n // 2) - 150000
    sc[32 * lo:32 * (lo + 300000)] = bytes(sc[0:32]) * 300000
    sc[32 * 1000:32 * 51000] = bytes(32 * 50000)
    sc = bytes(sc)
    want = common.oracle_msm(sc, pt, n)
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    assert mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == want
    assert mx.last_msm_shape()[0] == 17
    # the same input from host buffers: 4 ranges on one device into one bucket array
    assert mx.msm_host_multi("bn254", sc, pt, n, shards=0, devices=1) == want
    assert mx.last_msm_multi() == (4, 1)
    from porla_amd import lib
    lib.porla_gpu_release_msm_workspaces()


def test_rccl_from_cxx_world_of_one(mx, inputs):
    """porla_dist_*: ncclGetUniqueId / ncclCommInitRank / ncclAllGather bound with dlopen and issued from C++ (a world of one
    rank here; the two-rank form runs wherever two devices are visible)"""
    import torch
    sc, pt = inputs
    n = 3000
    d_sc = torch.frombuffer(bytearray(sc[:32 * n]), dtype=torch.uint8).cuda()
    d_pt = torch.frombuffer(bytearray(pt[:64 * n]), dtype=torch.uint8).cuda()
    uid = mx.dist_unique_id()
    mx.dist_init(uid, 0, 1)
    try:
        assert mx.dist_info() == (0, 1)
        want = common.oracle_msm(sc, pt, n)
        s = torch.cuda.current_stream().cuda_stream
        assert mx.msm_device_dist("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s) == want
        part = mx.msm_device("bn254", d_sc.data_ptr(), d_pt.data_ptr(), n, s, partial=True)
        assert mx.dist_allgather_partials(part, 1) == part
        assert mx.dist_fold("bn254", part) == want
    finally:
        mx.dist_finalize()
    assert mx.dist_info()[1] == 0


RANK_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["PORLA_ROOT"])
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=world)      # only to hand the ncclUniqueId around
from porla_amd import multiexp as mx
from tests import common
uid = [mx.dist_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
mx.dist_init(uid[0], rank, world)
n = 1 << 14
sc, pt = common.synth_inputs(n)
lo, hi = n * rank // world, n * (rank + 1) // world
d_sc = torch.frombuffer(bytearray(sc[32 * lo:32 * hi]), dtype=torch.uint8).cuda()
d_pt = torch.frombuffer(bytearray(pt[64 * lo:64 * hi]), dtype=torch.uint8).cuda()
got = mx.msm_device_dist("bn254", d_sc.data_ptr(), d_pt.data_ptr(), hi - lo, torch.cuda.current_stream().cuda_stream)
assert got == common.oracle_msm(sc, pt, n), got.hex()
mx.dist_finalize()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_rccl_from_cxx_two_ranks():
    """the real nccl branch: two processes, two devices, range partials over ncclAllGather -- skipped on one-GPU boxes"""
    if device_count() < 2:
        pytest.skip("needs two visible devices")
    common.synth_inputs(1 << 14)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PORLA_ROOT=common.ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", RANK_WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o


TIMEOUT_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["PORLA_ROOT"])
import torch
torch.cuda.set_device(0)
from porla_amd import multiexp as mx
uid = mx.dist_unique_id()
msgs = []
for attempt in range(2):
    try:
        mx.dist_init(uid, 0, 2)              # rank 0 of a world of two whose other rank never comes
        msgs.append("unexpected success")
    except Exception as e:
        msgs.append(str(e))
try:
    mx.dist_fold("bn254", bytes(96))
    msgs.append("unexpected success")
except Exception as e:
    msgs.append(str(e))
print("\n".join("MSG " + m.replace("\n", " ") for m in msgs), flush=True)
os._exit(7)                                  # what the header prescribes after a timed-out initialisation
"""


def test_dist_init_times_out_poisons_the_state_and_the_process_leaves():
    """porla_dist_init with a peer that never arrives (advisor r4): bounded by PORLA_DIST_INIT_TIMEOUT_S, after which the helper
    thread inside ncclCommInitRank is abandoned, a retry with the same id is REFUSED (no second helper on one communicator), the
    collectives refuse too, and the process leaves with _exit"""
    env = dict(os.environ, PORLA_ROOT=common.ROOT, PORLA_DIST_INIT_TIMEOUT_S="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", TIMEOUT_WORKER], env=env, capture_output=True, text=True, timeout=300)
    lines = [l[4:] for l in r.stdout.splitlines() if l.startswith("MSG ")]      # (RCCL prints its version banner on stdout)
    assert r.returncode == 7, r.stdout[-1500:] + r.stderr[-1500:]
    assert len(lines) == 3 and "did not return within 4 s" in lines[0], lines
    assert "timed out" in lines[1] and "fresh process" in lines[1], lines
    assert "timed out" in lines[2], lines
    assert time.time() - t0 < 120
