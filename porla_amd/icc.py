"""Host-side mirror of the ICC encode step of the reference's Server (CRebuild_Cached data part,
porla/Server/Server.hpp:1487-1833, and the scalar part of align_MAC, Server.hpp:531-541) over the engine's
C ABI (include/porla_gpu.h: porla_icc_encode_*).  Rows are in the reference's own formats: 32-byte
little-endian chunks in (utils.h:353-364), 64-byte little-endian values mod LCM out (utils.h:473-517)."""
import ctypes

from .loader import lib

CURVE = {"bn254": 0, "secp256k1": 1}
NUM_CHUNKS = 128  # config.hpp:22


def _check(rc):
    if rc != 0:
        raise RuntimeError("porla engine error %d: %s" % (rc, lib.porla_gpu_last_error().decode()))


def crebuild_host(rows, n_rows, n_cols=NUM_CHUNKS, curve="bn254", write_step=0, part=0, want_x=True,
                  want_aligned=True, want_scalars=True, scalar_le=False):
    """rows: n_rows*n_cols*32 bytes.  Returns (x_rows, aligned_rows, scalars) as bytes (or None)."""
    total = n_rows * n_cols
    x = ctypes.create_string_buffer(64 * total) if want_x else None
    al = ctypes.create_string_buffer(32 * total) if want_aligned else None
    sc = ctypes.create_string_buffer(32 * total) if want_scalars else None
    _check(lib.porla_icc_encode_host(bytes(rows), n_rows, n_cols, CURVE[curve], write_step, part,
                                     ctypes.cast(x, ctypes.c_void_p) if x else None,
                                     ctypes.cast(al, ctypes.c_void_p) if al else None,
                                     ctypes.cast(sc, ctypes.c_void_p) if sc else None, 1 if scalar_le else 0))
    return (x.raw if x else None, al.raw if al else None, sc.raw if sc else None)


def crebuild_cols_host(rows, n_rows, n_cols, col_begin, col_end, x, aligned, scalars, curve="bn254", write_step=0, part=0,
                       scalar_le=False):
    """column sharding: encode columns [col_begin, col_end) of the row-major `rows` into the same columns of the caller's
    full-width ctypes buffers x / aligned / scalars (each may be None)"""
    vp = ctypes.c_void_p
    _check(lib.porla_icc_encode_cols_host(bytes(rows), n_rows, n_cols, col_begin, col_end, CURVE[curve], write_step, part,
                                          ctypes.cast(x, vp) if x else None, ctypes.cast(aligned, vp) if aligned else None,
                                          ctypes.cast(scalars, vp) if scalars else None, 1 if scalar_le else 0))


def crebuild_host_multi(rows, n_rows, n_cols=NUM_CHUNKS, curve="bn254", write_step=0, part=0, devices=0, scalar_le=False):
    """the whole encode with the columns split over `devices` GPUs of this process (0 = every visible one)"""
    total = n_rows * n_cols
    x, al, sc = (ctypes.create_string_buffer(64 * total), ctypes.create_string_buffer(32 * total), ctypes.create_string_buffer(32 * total))
    vp = ctypes.c_void_p
    _check(lib.porla_icc_encode_host_multi(bytes(rows), n_rows, n_cols, CURVE[curve], write_step, part, ctypes.cast(x, vp),
                                           ctypes.cast(al, vp), ctypes.cast(sc, vp), 1 if scalar_le else 0, devices))
    return x.raw, al.raw, sc.raw


def crebuild_device(d_rows, n_rows, n_cols, curve, write_step, part, d_x=0, d_aligned=0, d_scalars=0, scalar_le=False,
                    stream=0):
    """device-pointer form (integers, e.g. torch tensor .data_ptr()); asynchronous on `stream`."""
    _check(lib.porla_icc_encode_device(ctypes.c_void_p(d_rows), n_rows, n_cols, CURVE[curve], write_step, part,
                                       ctypes.c_void_p(d_x or None), ctypes.c_void_p(d_aligned or None),
                                       ctypes.c_void_p(d_scalars or None), 1 if scalar_le else 0, ctypes.c_void_p(stream)))


def crebuild_xy_device(d_rows, n_rows, n_cols, curve, write_step, d_x=0, d_aligned=0, d_scalars=0, d_y_x=0, d_y_aligned=0,
                       d_y_scalars=0, scalar_le=False, stream=0):
    """X and Y parts from one run of the network (Y_k = wt X_k mod LCM); device pointers, any output may be 0"""
    vp = ctypes.c_void_p
    _check(lib.porla_icc_encode_xy_device(vp(d_rows), n_rows, n_cols, CURVE[curve], write_step, vp(d_x or None), vp(d_aligned or None),
                                          vp(d_scalars or None), vp(d_y_x or None), vp(d_y_aligned or None), vp(d_y_scalars or None),
                                          1 if scalar_le else 0, vp(stream)))


def mac_crebuild_host(macs, n_rows, curve="bn254", write_step=0, part=0):
    """MAC halves of CRebuild_Cached (Server.hpp:1523-1536, 1590-1609, 1658-1676): n_rows 64-byte affine MACs in/out."""
    out = ctypes.create_string_buffer(64 * n_rows)
    _check(lib.porla_icc_mac_encode_host(bytes(macs), n_rows, CURVE[curve], write_step, part, out))
    return out.raw


def mac_crebuild_xy_host(macs, n_rows, curve="bn254", write_step=0):
    """both MAC halves of CRebuild_Cached from one butterfly network (Y_k = wt * X_k): (X part, Y part)"""
    ox, oy = ctypes.create_string_buffer(64 * n_rows), ctypes.create_string_buffer(64 * n_rows)
    _check(lib.porla_icc_mac_encode_xy_host(bytes(macs), n_rows, CURVE[curve], write_step, ox, oy))
    return ox.raw, oy.raw


def mac_crebuild_xy_device(d_macs, n_rows, curve, write_step, d_out_x, d_out_y, stream=0):
    _check(lib.porla_icc_mac_encode_xy_device(ctypes.c_void_p(d_macs), n_rows, CURVE[curve], write_step, ctypes.c_void_p(d_out_x),
                                              ctypes.c_void_p(d_out_y), ctypes.c_void_p(stream)))


def kzg_crebuild_stage_device(d_rows, n_rows, write_step, d_aligned_x, d_aligned_y, d_scalars_xy, d_commits_xy, d_macs, d_macs_x,
                              d_macs_y, stream=0):
    """the last encode stage of a CRebuild (KZG build) in one call: both parts' encode + alignment scalars + commitments and, beside
    them on a second stream inside, both MAC encodes (include/porla_gpu.h: porla_kzg_crebuild_stage_device)"""
    vp = ctypes.c_void_p
    _check(lib.porla_kzg_crebuild_stage_device(vp(d_rows), n_rows, ctypes.c_ulonglong(write_step), vp(d_aligned_x or None),
                                               vp(d_aligned_y or None), vp(d_scalars_xy), vp(d_commits_xy), vp(d_macs), vp(d_macs_x),
                                               vp(d_macs_y), vp(stream)))


def mac_crebuild_device(d_macs, n_rows, curve, write_step, part, d_out, stream=0):
    _check(lib.porla_icc_mac_encode_device(ctypes.c_void_p(d_macs), n_rows, CURVE[curve], write_step, part,
                                           ctypes.c_void_p(d_out), ctypes.c_void_p(stream)))


def audit_combine_device(d_rows64, d_idx64, d_coef64, n64, d_rows32, d_idx32, d_coef32, n32, n_cols, curve,
                         d_exact=0, d_aligned=0, d_aligned_be=0, d_scalars=0, stream=0):
    """Server::audit row combine + align_MAC scalar part (Server.hpp:790-828, 531-541); all arguments device addresses."""
    vp = ctypes.c_void_p
    _check(lib.porla_audit_combine_device(vp(d_rows64 or None), vp(d_idx64 or None), vp(d_coef64 or None), n64,
                                          vp(d_rows32 or None), vp(d_idx32 or None), vp(d_coef32 or None), n32, n_cols,
                                          CURVE[curve], vp(d_exact or None), vp(d_aligned or None), vp(d_aligned_be or None),
                                          vp(d_scalars or None), vp(stream)))


def mix_host(a0, a1, length, n_cols, n_total, curve="bn254"):
    """Server::mix data part (Server.hpp:1269-1278): two blocks of `length` rows of 64-byte symbols -> 2*length rows."""
    out = ctypes.create_string_buffer(2 * length * n_cols * 64)
    _check(lib.porla_icc_mix_host(bytes(a0), bytes(a1), length, n_cols, n_total, CURVE[curve], out))
    return out.raw


def mac_mix_host(a0, a1, length, n_total, curve="bn254"):
    """Server::mix MAC part (Server.hpp:1281-1318): two blocks of `length` 64-byte affine MACs -> 2*length MACs."""
    out = ctypes.create_string_buffer(2 * length * 64)
    _check(lib.porla_icc_mac_mix_host(bytes(a0), bytes(a1), length, n_total, CURVE[curve], out))
    return out.raw


# ---- Server::HAdd / Client::HAdd and the HRebuild chains (Server.hpp:1388-1477, 1329-1386; Client.hpp:978-1038) ----
def hadd_host(data, n_total, write_step, curve="bn254", n_cols=NUM_CHUNKS, scalar_le=False):
    """data side of HAdd on one block: (data_B2 aligned mod p_icc [32-B LE each], alignment scalars, wt as 32-byte BE scalar)"""
    b2, sc, wt = ctypes.create_string_buffer(32 * n_cols), ctypes.create_string_buffer(32 * n_cols), ctypes.create_string_buffer(32)
    _check(lib.porla_icc_hadd_host(bytes(data), n_cols, n_total, write_step, CURVE[curve], b2, sc, 1 if scalar_le else 0, wt))
    return b2.raw, sc.raw, wt.raw


def mac_scale_host(mac, n_total, write_step, curve="bn254"):
    """MAC_B2 = wt * MAC (Server.hpp:1400-1417; Client::HAdd, Client.hpp:996-1014)"""
    out = ctypes.create_string_buffer(64)
    _check(lib.porla_icc_mac_scale_host(bytes(mac), n_total, write_step, CURVE[curve], out))
    return out.raw


def kzg_hadd_host(data, mac, n_total, write_step, n_cols=NUM_CHUNKS):
    """Server::HAdd for the KZG build: (data_B2, MAC_B2, MAC_align_B2)"""
    b2, m2, ma = ctypes.create_string_buffer(32 * n_cols), ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    _check(lib.porla_kzg_hadd_host(bytes(data), bytes(mac), n_total, write_step, b2, m2, ma))
    return b2.raw, m2.raw, ma.raw


def _level_ptrs(bufs):
    arr = (ctypes.c_void_p * len(bufs))()
    for i, b in enumerate(bufs):
        arr[i] = ctypes.cast(b, ctypes.c_void_p)
    return arr


def hrebuild_host(level_bufs, level, n_total, curve="bn254", n_cols=NUM_CHUNKS):
    """level_bufs[i]: ctypes buffer of 2 * 2^i rows x n_cols x 64 bytes (resident half, incoming half); rebuilt in place"""
    _check(lib.porla_icc_hrebuild_host(_level_ptrs(level_bufs), level, n_cols, n_total, CURVE[curve]))


def mac_hrebuild_host(level_bufs, level, n_total, curve="bn254"):
    """the same for 64-byte affine points (MAC commitments, MAC alignments, the client's complements)"""
    _check(lib.porla_icc_mac_hrebuild_host(_level_ptrs(level_bufs), level, n_total, CURVE[curve]))
