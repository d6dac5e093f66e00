"""Host-side mirror of the reference's KZG wrapper layer (porla/Utils/utils.h:235-305) over libmultiexp.so.

Same names and argument meaning as the C++ wrappers the reference's Server/Client call:
  bn254_add / bn254_mult / bn254_neg / bn254_set_infinity      utils.h:235-269
  bn254_scalar_set_int                                          utils.h:271-275
  bn254_multi_exp                                               utils.h:277-292
  bn254_compare                                                 utils.h:294-305
plus the direct GoSlice calls (init_key, init_SRS, ... Client.hpp:159-167,348-354,411-419,445-453,1637-1662;
Server.hpp:183-188,365-397,550-558).  Buffers are Python bytes/bytearray; results are returned as bytes.
"""
import ctypes

from .loader import GoSlice, lib

MAC_SIZE = 64       # COMMITMENT_MAC_SIZE with ENABLE_KZG, config.hpp:26
SCALAR_SIZE = 32    # bn254_scalar = uint32_t[8], utils.h:64


def _slice(buf):
    """GoSlice over a mutable ctypes buffer."""
    return GoSlice(ctypes.cast(buf, ctypes.c_void_p), len(buf), len(buf))


def _buf(data):
    return ctypes.create_string_buffer(bytes(data), len(data))


def _check(rc):
    if rc != 0:
        raise RuntimeError("porla engine error %d: %s" % (rc, lib.porla_gpu_last_error().decode()))


# ---- utils.h wrappers ---------------------------------------------------------------------------
def bn254_add(a, b):
    """utils.h:235-244 -> add_point (in place on a; returned here)."""
    ba, bb = _buf(a), _buf(b)
    sa, sb = _slice(ba), _slice(bb)
    lib.add_point(ctypes.byref(sa), ctypes.byref(sb))
    return ba.raw


def bn254_mult(a, scalar):
    """utils.h:246-255 -> mult_point."""
    ba, bs = _buf(a), _buf(scalar)
    sa, ss = _slice(ba), _slice(bs)
    lib.mult_point(ctypes.byref(sa), ctypes.byref(ss))
    return ba.raw


def bn254_neg(a):
    """utils.h:257-262 -> neg_point."""
    ba = _buf(a)
    sa = _slice(ba)
    lib.neg_point(ctypes.byref(sa))
    return ba.raw


def bn254_set_infinity():
    """utils.h:264-269 -> set_inf_point."""
    ba = _buf(b"\xff" * MAC_SIZE)
    sa = _slice(ba)
    lib.set_inf_point(ctypes.byref(sa))
    return ba.raw


def bn254_scalar_set_int(v):
    """utils.h:271-275: 28 zero bytes then v big-endian."""
    return bytes(28) + int(v & 0xffffffff).to_bytes(4, "big")


def bn254_multi_exp(points, scalars, n):
    """utils.h:277-292 -> compute_multi_exp(scalars, points, n, result)."""
    bs, bp, out = _buf(scalars), _buf(points), _buf(bytes(MAC_SIZE))
    ss, sp, so = _slice(bs), _slice(bp), _slice(out)
    lib.compute_multi_exp(ctypes.byref(ss), ctypes.byref(sp), n, ctypes.byref(so))
    return out.raw


def bn254_compare(a, b):
    """utils.h:294-305 -> compare_commitment."""
    ba, bb = _buf(a), _buf(b)
    sa, sb = _slice(ba), _slice(bb)
    return bool(lib.compare_commitment(ctypes.byref(sa), ctypes.byref(sb)))


# ---- direct cgo calls ---------------------------------------------------------------------------
def init_key(tau, alpha):
    bt, ba = _buf(tau), _buf(alpha)
    st, sa = _slice(bt), _slice(ba)
    lib.init_key(ctypes.byref(st), ctypes.byref(sa))


def init_SRS(n):
    """Client.hpp:348-354: returns the 32n+132-byte wire blob."""
    out = _buf(bytes(32 * n + 132 + 64))
    so = _slice(out)
    ln = ctypes.c_longlong(0)
    lib.init_SRS(n, ctypes.byref(so), ctypes.byref(ln))
    return out.raw[:ln.value]


def init_SRS_from_data(n, blob):
    bb = _buf(blob)
    sb = _slice(bb)
    lib.init_SRS_from_data(n, ctypes.byref(sb))


def _in_out(fn, data, out_len):
    bi, bo = _buf(data), _buf(bytes(out_len))
    si, so = _slice(bi), _slice(bo)
    fn(ctypes.byref(si), ctypes.byref(so))
    return bo.raw


def compute_digest(data):
    return _in_out(lib.compute_digest, data, MAC_SIZE)


def compute_digest_complement(data):
    return _in_out(lib.compute_digest_complement, data, MAC_SIZE)


def compute_digest_from_srs(data):
    return _in_out(lib.compute_digest_from_srs, data, MAC_SIZE)


def create_proof(random_point, data):
    """Server.hpp:363-398: returns (commitment 64, H 64, point 32, claim 32)."""
    bi = _buf(data)
    outs = [_buf(bytes(64)), _buf(bytes(64)), _buf(bytes(32)), _buf(bytes(32))]
    si = _slice(bi)
    so = [_slice(o) for o in outs]
    lib.create_proof(random_point, ctypes.byref(si), *[ctypes.byref(s) for s in so])
    return tuple(o.raw for o in outs)


def verify_proof(commitment, proof_h, point, claim):
    bufs = [_buf(commitment), _buf(proof_h), _buf(point), _buf(claim)]
    sl = [_slice(b) for b in bufs]
    return bool(lib.verify_proof(*[ctypes.byref(s) for s in sl]))


# ---- device-pointer API (include/porla_gpu.h) -------------------------------------------------------
def msm_device(curve, d_scalars, d_points, n, stream=0, partial=False):
    """d_scalars / d_points: integer device addresses (e.g. torch tensor .data_ptr())."""
    out = ctypes.create_string_buffer(96 if partial else 64)
    fn = getattr(lib, "porla_%s_msm_device%s" % (curve, "_partial" if partial else ""))
    _check(fn(ctypes.c_void_p(d_scalars), ctypes.c_void_p(d_points), n, out, ctypes.c_void_p(stream)))
    return out.raw


def msm_begin(slot, d_scalars, d_points, n, stream=0, curve="bn254"):
    """two-phase MSM: enqueue on `stream` into workspace slot 1..3 (include/porla_gpu.h)"""
    fn = getattr(lib, "porla_%s_msm_device_begin" % curve)
    _check(fn(slot, ctypes.c_void_p(d_scalars), ctypes.c_void_p(d_points), n, ctypes.c_void_p(stream)))


def msm_end(slot, partial=False, curve="bn254"):
    out = ctypes.create_string_buffer(96 if partial else 64)
    _check(getattr(lib, "porla_%s_msm_device_end" % curve)(slot, out, 1 if partial else 0))
    return out.raw


def msm_host(curve, scalars, points, n):
    out = ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_msm_host" % curve)(bytes(scalars), bytes(points), n, out))
    return out.raw


def msm_pair_device(curve, d_scalars, d_points_a, d_points_b, n, stream=0):
    """the audit's two MSMs over one scalar array (Server.hpp:900-901 / :842-848) in one call: (sum s_i A_i, sum s_i B_i)"""
    oa, ob = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_msm_pair_device" % curve)(ctypes.c_void_p(d_scalars), ctypes.c_void_p(d_points_a), ctypes.c_void_p(d_points_b),
                                                            n, oa, ob, ctypes.c_void_p(stream)))
    return oa.raw, ob.raw


def audit_msm_pair_device(curve, d_store_a, d_store_b, d_idx, d_coef, n, stream=0):
    """the audit's two MSMs from resident MAC arrays: (sum coef_i store_a[idx_i], sum coef_i store_b[idx_i]); idx int64, coef abs(int32)"""
    oa, ob = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    vp = ctypes.c_void_p
    _check(getattr(lib, "porla_%s_audit_msm_pair_device" % curve)(vp(d_store_a), vp(d_store_b), vp(d_idx), vp(d_coef), n, oa, ob, vp(stream)))
    return oa.raw, ob.raw


def audit_msm_pair_begin(slot, curve, d_store_a, d_store_b, d_idx, d_coef, n, stream=0):
    vp = ctypes.c_void_p
    _check(getattr(lib, "porla_%s_audit_msm_pair_begin" % curve)(slot, vp(d_store_a), vp(d_store_b), vp(d_idx), vp(d_coef), n, vp(stream)))


def audit_msm_pair_end(slot, curve):
    oa, ob = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_audit_msm_pair_end" % curve)(slot, oa, ob))
    return oa.raw, ob.raw


def msm_pair_host(curve, scalars, points_a, points_b, n):
    oa, ob = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_msm_pair_host" % curve)(bytes(scalars), bytes(points_a), bytes(points_b), n, oa, ob))
    return oa.raw, ob.raw


def msm_host_multi(curve, scalars, points, n, shards=0, devices=0):
    """range-sharded over `shards` pair ranges and `devices` GPUs of this process (0 = automatic), behind the C ABI"""
    out = ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_msm_host_multi" % curve)(bytes(scalars), bytes(points), n, shards, devices, out))
    return out.raw


def last_msm_multi():
    """(ranges, devices) the most recent msm_host_multi used"""
    s, d = ctypes.c_int(0), ctypes.c_int(0)
    lib.porla_gpu_last_msm_multi(ctypes.byref(s), ctypes.byref(d))
    return s.value, d.value


# ---- one process per GPU: RCCL all-gather of the 96-byte partials, issued from C++ (include/porla_gpu.h) ----
def dist_unique_id():
    out = ctypes.create_string_buffer(128)
    _check(lib.porla_dist_unique_id(out))
    return out.raw


def dist_init(unique_id, rank, world):
    _check(lib.porla_dist_init(bytes(unique_id), rank, world))


def dist_info():
    r, w = ctypes.c_int(0), ctypes.c_int(0)
    lib.porla_dist_info(ctypes.byref(r), ctypes.byref(w))
    return r.value, w.value


def dist_finalize():
    _check(lib.porla_dist_finalize())


def dist_allgather_partials(partial, world):
    out = ctypes.create_string_buffer(96 * world)
    _check(lib.porla_dist_allgather_partials(bytes(partial), out))
    return out.raw


def dist_fold(curve, partial):
    """all ranks' 96-byte partials through one ncclAllGather, folded: the whole job's 64-byte result"""
    out = ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_dist_fold" % curve)(bytes(partial), out))
    return out.raw


def msm_device_dist(curve, d_scalars, d_points, n_local, stream=0):
    out = ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_msm_device_dist" % curve)(ctypes.c_void_p(d_scalars), ctypes.c_void_p(d_points), n_local, out,
                                                            ctypes.c_void_p(stream)))
    return out.raw


def jac_sum(curve, jacobians, count):
    out = ctypes.create_string_buffer(64)
    _check(getattr(lib, "porla_%s_jac_sum" % curve)(bytes(jacobians), count, out))
    return out.raw


# ---- batched fixed-base commitments (include/porla_gpu.h) ------------------------------------------
CURVES = {"bn254": 0, "secp256k1": 1}


class FixedBase:
    """Resident window-multiples table of a fixed base; commit() = compute_digest_from_srs / compute_commitment
    hoisted over many rows (Server.hpp:550-560, Client.hpp:374-406)."""

    def __init__(self, curve, points, n_points, window_bits=0):
        self.h = ctypes.c_void_p()
        _check(lib.porla_fixed_base_create(CURVES[curve], bytes(points), n_points, window_bits, ctypes.byref(self.h)))

    def info(self):
        c, w, b = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_ulonglong(0)
        _check(lib.porla_fixed_base_info(self.h, ctypes.byref(c), ctypes.byref(w), ctypes.byref(b)))
        return {"window_bits": c.value, "windows": w.value, "table_bytes": b.value}

    def commit_host(self, rows, n_rows, n_coeffs, row_stride=None):
        out = ctypes.create_string_buffer(64 * max(n_rows, 1))
        _check(lib.porla_fixed_base_commit_host(self.h, bytes(rows), n_rows, n_coeffs, row_stride or 32 * n_coeffs, out))
        return out.raw[:64 * n_rows]

    def commit_device(self, d_rows, n_rows, n_coeffs, d_out, stream=0, row_stride=None):
        _check(lib.porla_fixed_base_commit_device(self.h, ctypes.c_void_p(d_rows), n_rows, n_coeffs,
                                                  row_stride or 32 * n_coeffs, ctypes.c_void_p(d_out),
                                                  ctypes.c_void_p(stream)))

    def ipa_audit_device(self, d_rows64, d_idx64, d_coef64, n64, d_rows32, d_idx32, d_coef32, n32, n_cols, d_mac_store, d_align_store,
                         d_mac_idx, d_mac_coef, n_macs, stream=0):
        """Server::audit (IPA build) up to the proof, self = the generators' fixed base ->
        dict(combined_mac, combined_align, align_value, commitment, b)"""
        vp = ctypes.c_void_p
        o = [ctypes.create_string_buffer(k) for k in (64, 64, 64, 64, 32 * n_cols)]
        _check(lib.porla_ipa_audit_device(self.h, vp(d_rows64 or None), vp(d_idx64 or None), vp(d_coef64 or None), n64,
                                          vp(d_rows32 or None), vp(d_idx32 or None), vp(d_coef32 or None), n32, n_cols, vp(d_mac_store),
                                          vp(d_align_store), vp(d_mac_idx), vp(d_mac_coef), n_macs, *o, vp(stream)))
        return dict(zip(("combined_mac", "combined_align", "align_value", "commitment", "b"), (x.raw for x in o)))

    def close(self):
        if self.h:
            lib.porla_fixed_base_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kzg_commit_batch_host(rows, n_rows):
    out = ctypes.create_string_buffer(64 * max(n_rows, 1))
    _check(lib.porla_kzg_commit_batch_host(bytes(rows), n_rows, out))
    return out.raw[:64 * n_rows]


def kzg_commit_batch_host_multi(rows, n_rows, devices=0):
    """row range split over `devices` GPUs of this process (0 = every visible one)"""
    out = ctypes.create_string_buffer(64 * max(n_rows, 1))
    _check(lib.porla_kzg_commit_batch_host_multi(bytes(rows), n_rows, out, devices))
    return out.raw[:64 * n_rows]


def shard_range(n, rank, world):
    """[begin, end) of shard `rank` of `world` over n units -- the engine's own range rule (porla_shard_range)"""
    b, e = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _check(lib.porla_shard_range(n, rank, world, ctypes.byref(b), ctypes.byref(e)))
    return b.value, e.value


def kzg_commit_batch_device(d_rows, n_rows, d_out, stream=0):
    _check(lib.porla_kzg_commit_batch_device(ctypes.c_void_p(d_rows), n_rows, ctypes.c_void_p(d_out), ctypes.c_void_p(stream)))


def kzg_commit_batch_device_to_host(d_rows, n_rows, stream=0):
    """commitments of rows resident on the device, results on the host (blocking; <= 64 rows: one launch)"""
    out = ctypes.create_string_buffer(64 * n_rows)
    _check(lib.porla_kzg_commit_batch_device_to_host(ctypes.c_void_p(d_rows), n_rows, out, ctypes.c_void_p(stream)))
    return out.raw


def kzg_audit_device(d_rows64, d_idx64, d_coef64, n64, d_rows32, d_idx32, d_coef32, n32, d_mac_store, d_align_store, d_mac_idx,
                     d_mac_coef, n_macs, z, n_cols=None, stream=0):
    """Server::audit (KZG) in one call -> dict(combined_mac, combined_align, align_value, commitment, proof_h, point, claim, b).
    The library writes 32 bytes per SRS coefficient into `b`: n_cols, if given, must be the SRS size"""
    vp = ctypes.c_void_p
    srs_n = kzg_row_coefficients()
    if n_cols is None:
        n_cols = srs_n
    elif n_cols != srs_n:
        raise ValueError("kzg_audit_device: n_cols=%d but the SRS holds %d coefficients" % (n_cols, srs_n))
    o = [ctypes.create_string_buffer(k) for k in (64, 64, 64, 64, 64, 32, 32, 32 * n_cols)]
    _check(lib.porla_kzg_audit_device(vp(d_rows64 or None), vp(d_idx64 or None), vp(d_coef64 or None), n64, vp(d_rows32 or None),
                                      vp(d_idx32 or None), vp(d_coef32 or None), n32, vp(d_mac_store), vp(d_align_store), vp(d_mac_idx),
                                      vp(d_mac_coef), n_macs, z, *o, vp(stream)))
    return dict(zip(("combined_mac", "combined_align", "align_value", "commitment", "proof_h", "point", "claim", "b"), (x.raw for x in o)))


def kzg_digest_batch_device(d_rows, n_rows, d_out, stream=0):
    """compute_digest (Client.hpp:408-419) over n_rows rows resident in HBM"""
    _check(lib.porla_kzg_digest_batch_device(ctypes.c_void_p(d_rows), n_rows, ctypes.c_void_p(d_out), ctypes.c_void_p(stream)))


def kzg_complement_batch_device(d_scalars, n, d_out, stream=0):
    """compute_digest_complement (Client.hpp:445-453) over n 32-byte big-endian scalars resident in HBM"""
    _check(lib.porla_kzg_complement_batch_device(ctypes.c_void_p(d_scalars), n, ctypes.c_void_p(d_out), ctypes.c_void_p(stream)))


def kzg_mac_batch_device(d_rows, d_scalars, n_rows, d_out, stream=0):
    """digest(row) + complement(scalar) per block, the MAC Client::initialize / update send (Client.hpp:229-236, 468-478)"""
    _check(lib.porla_kzg_mac_batch_device(ctypes.c_void_p(d_rows), ctypes.c_void_p(d_scalars), n_rows, ctypes.c_void_p(d_out),
                                          ctypes.c_void_p(stream)))


def kzg_digest_batch_host(rows, n_rows):
    """compute_digest over n_rows rows in host memory -> n_rows * 64 bytes"""
    out = ctypes.create_string_buffer(64 * n_rows)
    _check(lib.porla_kzg_digest_batch_host(rows, n_rows, out))
    return out.raw


def kzg_complement_batch_host(scalars, n):
    """compute_digest_complement over n 32-byte big-endian scalars in host memory -> n * 64 bytes"""
    out = ctypes.create_string_buffer(64 * n)
    _check(lib.porla_kzg_complement_batch_host(scalars, n, out))
    return out.raw


def kzg_mac_batch_host(rows, scalars, n_rows):
    """digest(row) + complement(scalar) per block, host buffers -> n_rows * 64 bytes"""
    out = ctypes.create_string_buffer(64 * n_rows)
    _check(lib.porla_kzg_mac_batch_host(rows, scalars, n_rows, out))
    return out.raw


def profile_enable(on=True):
    """False/0: off; True/1: HIP events around every kernel; 2: around the workload's dominant kernel only"""
    lib.porla_gpu_profile_enable(int(on))


def kzg_row_coefficients():
    """coefficients per commitment row = the SRS size of this process (0 before init_SRS / init_SRS_from_data)"""
    n = ctypes.c_size_t(0)
    _check(lib.porla_kzg_row_coefficients(ctypes.byref(n)))
    return n.value


def kzg_commit_shape():
    """(window bits, windows per coefficient) of the resident SRS table"""
    c, w = ctypes.c_int(0), ctypes.c_int(0)
    lib.porla_kzg_commit_shape(ctypes.byref(c), ctypes.byref(w))
    return c.value, w.value


def last_msm_shape():
    """(window bits, window count, GLV flag) of the most recently launched MSM"""
    c, w, g = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    lib.porla_gpu_last_msm_shape(ctypes.byref(c), ctypes.byref(w), ctypes.byref(g))
    return c.value, w.value, bool(g.value)


def profile_get():
    """[(kernel name, total ms, launches)]"""
    res = []
    i = 0
    while True:
        name = ctypes.create_string_buffer(64)
        ms = ctypes.c_double(0)
        cnt = ctypes.c_longlong(0)
        if lib.porla_gpu_profile_get(i, name, 64, ctypes.byref(ms), ctypes.byref(cnt)) != 0:
            break
        res.append((name.value.decode(), ms.value, cnt.value))
        i += 1
    return res
