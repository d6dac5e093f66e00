"""Multi-GPU form of the MSM: input-pair range sharding, one process per GPU.

Each rank runs a full bucket MSM over its own pairs and produces ONE partial sum (96-byte Jacobian).  The only
exchange step is an all_gather of world_size x 96 bytes (RCCL over xGMI with backend "nccl"; gloo on CPU in the
tests) followed by world_size - 1 group additions and one inversion -- the same fold the reference performs across
its 8 pool threads (porla/Client/Client.hpp:761-787).  EC addition is not an RCCL reduce op, hence gather + fold."""
import torch
import torch.distributed as dist

from . import multiexp as mx

PARTIAL_BYTES = 96


def gather_partials(partial, device=None):
    """all_gather of this rank's 96-byte partial; returns the list of all ranks' partials (bytes)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [bytes(partial)]
    mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    out = [torch.zeros(PARTIAL_BYTES, dtype=torch.uint8, device=mine.device) for _ in range(world)]
    dist.all_gather(out, mine)
    return [bytes(t.cpu().numpy().tobytes()) for t in out]


def fold_partials(curve, partials):
    """sum of Jacobian partials -> 64-byte affine result (host: latency-bound, <= 7 additions)"""
    return mx.jac_sum(curve, b"".join(partials), len(partials))


def sharded_msm_device(curve, d_scalars, d_points, n_local, stream=0, device=None):
    """this rank's pairs are resident at d_scalars / d_points; returns the whole-job 64-byte affine result."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return mx.msm_device(curve, d_scalars, d_points, n_local, stream)
    part = mx.msm_device(curve, d_scalars, d_points, n_local, stream, partial=True)
    return fold_partials(curve, gather_partials(part, device))


def affine_to_partial(affine64):
    """64-byte affine -> 96-byte Jacobian with Z = 1 (infinity -> Z = 0)"""
    if bytes(affine64) == bytes(64):
        return (1).to_bytes(32, "big") * 2 + bytes(32)
    return bytes(affine64) + (1).to_bytes(32, "big")


# ---- the paths that shard with NO collective: every rank works on its own range and keeps its results
def my_range(n):
    """this rank's [begin, end) of n units (porla_shard_range: rows of a commitment batch, columns of an ICC encode)"""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    return mx.shard_range(n, rank, world)


def sharded_commit_rows(rows, n_rows, commit=None, row_coefficients=None):
    """this rank's row range of a commitment batch -> (first row, its 64-byte commitments).  `commit(rows, n)` defaults to the
    engine (porla_kzg_commit_batch_host); the CPU tests pass the oracle in its place.  A row is `row_coefficients` 32-byte
    coefficients: by default the library's SRS size (porla_kzg_row_coefficients; the reference's NUM_CHUNKS = 128), or --
    with a stand-in `commit` and no count given -- len(rows) / n_rows."""
    lo, hi = my_range(n_rows)
    fn = commit or mx.kzg_commit_batch_host
    if row_coefficients is None:
        row_coefficients = mx.kzg_row_coefficients() if commit is None else (len(rows) // (32 * n_rows) if n_rows else 0)
    stride = 32 * row_coefficients
    if n_rows and (stride == 0 or len(rows) != stride * n_rows):
        raise ValueError("commitment rows: %d bytes for %d rows of %d coefficients" % (len(rows), n_rows, row_coefficients))
    return lo, fn(rows[stride * lo:stride * hi], hi - lo)


def gather_objects(obj):
    """test / bookkeeping helper: every rank's object (NOT part of the data path -- the results stay where they are produced)"""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
