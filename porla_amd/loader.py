"""ctypes loader for libmultiexp.so (fails loudly when the HIP extension is not built)."""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "libmultiexp.so")
_lib = None


class GoSlice(ctypes.Structure):
    """cgo slice header, porla/Utils/libmultiexp.h:61."""
    _fields_ = [("data", ctypes.c_void_p), ("len", ctypes.c_longlong), ("cap", ctypes.c_longlong)]


def load():
    """Load the engine.  When torch is importable it is imported FIRST so that the HIP runtime the process
    ends up with is the one torch ships (both have soname libamdhip64.so.7; two runtimes in one process
    cannot share device pointers)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(lib_path):
        raise ImportError(
            "porla_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C porla_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % lib_path)
    if os.environ.get("PORLA_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    _lib = ctypes.CDLL(lib_path, mode=ctypes.RTLD_GLOBAL)
    _declare(_lib)
    return _lib


def _declare(L):
    P = ctypes.POINTER(GoSlice)
    vp, sz, u8p = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p
    L.init_key.argtypes = [P, P]; L.init_key.restype = None
    L.init_SRS.argtypes = [ctypes.c_longlong, P, ctypes.POINTER(ctypes.c_longlong)]; L.init_SRS.restype = None
    L.init_SRS_from_data.argtypes = [ctypes.c_longlong, P]; L.init_SRS_from_data.restype = None
    for name in ("compute_digest", "compute_digest_complement", "compute_digest_from_srs", "add_point", "mult_point"):
        getattr(L, name).argtypes = [P, P]; getattr(L, name).restype = None
    L.compute_multi_exp.argtypes = [P, P, ctypes.c_longlong, P]; L.compute_multi_exp.restype = None
    L.compare_commitment.argtypes = [P, P]; L.compare_commitment.restype = ctypes.c_ubyte
    L.create_proof.argtypes = [ctypes.c_ulonglong, P, P, P, P, P]; L.create_proof.restype = None
    L.verify_proof.argtypes = [P, P, P, P]; L.verify_proof.restype = ctypes.c_ubyte
    L.neg_point.argtypes = [P]; L.neg_point.restype = None
    L.set_inf_point.argtypes = [P]; L.set_inf_point.restype = None

    L.porla_gpu_device_count.argtypes = []; L.porla_gpu_device_count.restype = ctypes.c_int
    L.porla_gpu_set_device.argtypes = [ctypes.c_int]; L.porla_gpu_set_device.restype = ctypes.c_int
    L.porla_gpu_last_error.argtypes = []; L.porla_gpu_last_error.restype = ctypes.c_char_p
    L.porla_gpu_profile_enable.argtypes = [ctypes.c_int]; L.porla_gpu_profile_enable.restype = ctypes.c_int
    L.porla_gpu_profile_get.argtypes = [ctypes.c_int, ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_longlong)]
    L.porla_gpu_profile_get.restype = ctypes.c_int
    L.porla_gpu_set_msm_window.argtypes = [ctypes.c_int]; L.porla_gpu_set_msm_window.restype = ctypes.c_int
    L.porla_gpu_set_msm_glv.argtypes = [ctypes.c_int]; L.porla_gpu_set_msm_glv.restype = ctypes.c_int
    L.porla_gpu_set_msm_small.argtypes = [ctypes.c_int, ctypes.c_int]; L.porla_gpu_set_msm_small.restype = ctypes.c_int
    L.porla_gpu_last_msm_shape.argtypes = [ctypes.POINTER(ctypes.c_int)] * 3; L.porla_gpu_last_msm_shape.restype = ctypes.c_int
    L.porla_glv_split.argtypes = [ctypes.c_int, u8p, u8p, ctypes.POINTER(ctypes.c_int), u8p, ctypes.POINTER(ctypes.c_int)]
    L.porla_glv_split.restype = ctypes.c_int
    L.porla_diag_fe_op.argtypes = [ctypes.c_int, ctypes.c_int, u8p, u8p, u8p]; L.porla_diag_fe_op.restype = ctypes.c_int
    for curve in ("bn254", "secp256k1"):
        f = getattr(L, "porla_%s_msm_device_begin" % curve); f.argtypes = [ctypes.c_int, vp, vp, sz, vp]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_msm_device_end" % curve); f.argtypes = [ctypes.c_int, u8p, ctypes.c_int]; f.restype = ctypes.c_int
    for curve in ("bn254", "secp256k1"):
        f = getattr(L, "porla_%s_msm_device" % curve); f.argtypes = [vp, vp, sz, u8p, vp]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_msm_device_partial" % curve); f.argtypes = [vp, vp, sz, u8p, vp]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_msm_host" % curve); f.argtypes = [u8p, u8p, sz, u8p]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_msm_pair_device" % curve); f.argtypes = [vp, vp, vp, sz, u8p, u8p, vp]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_audit_msm_pair_device" % curve); f.argtypes = [vp, vp, vp, vp, sz, u8p, u8p, vp]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_audit_msm_pair_begin" % curve); f.argtypes = [ctypes.c_int, vp, vp, vp, vp, sz, vp]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_audit_msm_pair_end" % curve); f.argtypes = [ctypes.c_int, u8p, u8p]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_msm_pair_host" % curve); f.argtypes = [u8p, u8p, u8p, sz, u8p, u8p]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_msm_host_multi" % curve); f.argtypes = [u8p, u8p, sz, ctypes.c_int, ctypes.c_int, u8p]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_dist_fold" % curve); f.argtypes = [u8p, u8p]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_msm_device_dist" % curve); f.argtypes = [vp, vp, sz, u8p, vp]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_jac_sum" % curve); f.argtypes = [u8p, sz, u8p]; f.restype = ctypes.c_int
        f = getattr(L, "porla_%s_tree_fold" % curve); f.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p]; f.restype = ctypes.c_int
    L.porla_icc_hadd_host.argtypes = [u8p, sz, sz, ctypes.c_ulonglong, ctypes.c_int, u8p, u8p, ctypes.c_int, u8p]; L.porla_icc_hadd_host.restype = ctypes.c_int
    L.porla_icc_mac_scale_host.argtypes = [u8p, sz, ctypes.c_ulonglong, ctypes.c_int, u8p]; L.porla_icc_mac_scale_host.restype = ctypes.c_int
    L.porla_kzg_hadd_host.argtypes = [u8p, u8p, sz, ctypes.c_ulonglong, u8p, u8p, u8p]; L.porla_kzg_hadd_host.restype = ctypes.c_int
    L.porla_icc_hrebuild_host.argtypes = [ctypes.POINTER(vp), ctypes.c_int, sz, sz, ctypes.c_int]; L.porla_icc_hrebuild_host.restype = ctypes.c_int
    L.porla_icc_mac_hrebuild_host.argtypes = [ctypes.POINTER(vp), ctypes.c_int, sz, ctypes.c_int]; L.porla_icc_mac_hrebuild_host.restype = ctypes.c_int
    L.porla_shard_range.argtypes = [sz, ctypes.c_int, ctypes.c_int, ctypes.POINTER(sz), ctypes.POINTER(sz)]; L.porla_shard_range.restype = ctypes.c_int
    L.porla_kzg_commit_batch_host_multi.argtypes = [u8p, sz, u8p, ctypes.c_int]; L.porla_kzg_commit_batch_host_multi.restype = ctypes.c_int
    L.porla_icc_encode_cols_host.argtypes = [u8p, sz, sz, sz, sz, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, vp, vp, vp, ctypes.c_int]
    L.porla_icc_encode_cols_host.restype = ctypes.c_int
    L.porla_icc_encode_host_multi.argtypes = [u8p, sz, sz, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, vp, vp, vp, ctypes.c_int, ctypes.c_int]
    L.porla_icc_encode_host_multi.restype = ctypes.c_int
    L.porla_gpu_last_msm_multi.argtypes = [ctypes.POINTER(ctypes.c_int)] * 2; L.porla_gpu_last_msm_multi.restype = ctypes.c_int
    L.porla_dist_unique_id.argtypes = [u8p]; L.porla_dist_unique_id.restype = ctypes.c_int
    L.porla_dist_init.argtypes = [u8p, ctypes.c_int, ctypes.c_int]; L.porla_dist_init.restype = ctypes.c_int
    L.porla_dist_info.argtypes = [ctypes.POINTER(ctypes.c_int)] * 2; L.porla_dist_info.restype = ctypes.c_int
    L.porla_dist_finalize.argtypes = []; L.porla_dist_finalize.restype = ctypes.c_int
    L.porla_dist_allgather_partials.argtypes = [u8p, u8p]; L.porla_dist_allgather_partials.restype = ctypes.c_int
    L.porla_fixed_base_create.argtypes = [ctypes.c_int, u8p, sz, ctypes.c_int, ctypes.POINTER(vp)]
    L.porla_fixed_base_create.restype = ctypes.c_int
    L.porla_fixed_base_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                        ctypes.POINTER(ctypes.c_ulonglong)]
    L.porla_fixed_base_info.restype = ctypes.c_int
    L.porla_fixed_base_commit_device.argtypes = [vp, vp, sz, sz, sz, vp, vp]; L.porla_fixed_base_commit_device.restype = ctypes.c_int
    L.porla_fixed_base_commit_host.argtypes = [vp, u8p, sz, sz, sz, u8p]; L.porla_fixed_base_commit_host.restype = ctypes.c_int
    L.porla_fixed_base_destroy.argtypes = [vp]; L.porla_fixed_base_destroy.restype = None
    L.porla_kzg_commit_batch_device.argtypes = [vp, sz, vp, vp]; L.porla_kzg_commit_batch_device.restype = ctypes.c_int
    L.porla_kzg_commit_batch_host.argtypes = [u8p, sz, u8p]; L.porla_kzg_commit_batch_host.restype = ctypes.c_int
    L.porla_kzg_commit_batch_device_to_host.argtypes = [vp, sz, u8p, vp]; L.porla_kzg_commit_batch_device_to_host.restype = ctypes.c_int
    L.porla_ipa_audit_device.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, sz, sz, vp, vp, vp, vp, sz] + [u8p] * 5 + [vp]
    L.porla_ipa_audit_device.restype = ctypes.c_int
    L.porla_kzg_audit_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, vp, vp, vp, vp, sz, ctypes.c_ulonglong] + [u8p] * 8 + [vp]
    L.porla_kzg_audit_device.restype = ctypes.c_int
    L.porla_kzg_digest_batch_device.argtypes = [vp, sz, vp, vp]; L.porla_kzg_digest_batch_device.restype = ctypes.c_int
    L.porla_kzg_complement_batch_device.argtypes = [vp, sz, vp, vp]; L.porla_kzg_complement_batch_device.restype = ctypes.c_int
    L.porla_kzg_mac_batch_device.argtypes = [vp, vp, sz, vp, vp]; L.porla_kzg_mac_batch_device.restype = ctypes.c_int
    L.porla_kzg_digest_batch_host.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p]; L.porla_kzg_digest_batch_host.restype = ctypes.c_int
    L.porla_kzg_complement_batch_host.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p]; L.porla_kzg_complement_batch_host.restype = ctypes.c_int
    L.porla_kzg_mac_batch_host.argtypes = [ctypes.c_char_p, ctypes.c_char_p, sz, ctypes.c_char_p]; L.porla_kzg_mac_batch_host.restype = ctypes.c_int
    L.porla_bn254_g2_mul_generator.argtypes = [u8p, u8p]; L.porla_bn254_g2_mul_generator.restype = ctypes.c_int
    L.porla_bn254_pairing_product_is_one.argtypes = [u8p, u8p, u8p, u8p, ctypes.c_int]; L.porla_bn254_pairing_product_is_one.restype = ctypes.c_int
    L.porla_kzg_set_commit_window.argtypes = [ctypes.c_int]; L.porla_kzg_set_commit_window.restype = ctypes.c_int
    L.porla_kzg_commit_shape.argtypes = [ctypes.POINTER(ctypes.c_int)] * 2; L.porla_kzg_commit_shape.restype = ctypes.c_int
    L.porla_kzg_row_coefficients.argtypes = [ctypes.POINTER(ctypes.c_size_t)]; L.porla_kzg_row_coefficients.restype = ctypes.c_int
    L.porla_kzg_release_device_memory.argtypes = []; L.porla_kzg_release_device_memory.restype = ctypes.c_int
    L.porla_gpu_release_msm_workspaces.argtypes = []; L.porla_gpu_release_msm_workspaces.restype = ctypes.c_int
    L.porla_icc_mac_encode_device.argtypes = [vp, sz, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, vp, vp]
    L.porla_icc_mac_encode_device.restype = ctypes.c_int
    L.porla_icc_encode_xy_device.argtypes = [vp, sz, sz, ctypes.c_int, ctypes.c_ulonglong, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp]
    L.porla_icc_encode_xy_device.restype = ctypes.c_int
    L.porla_icc_mac_encode_xy_device.argtypes = [vp, sz, ctypes.c_int, ctypes.c_ulonglong, vp, vp, vp]
    L.porla_icc_mac_encode_xy_device.restype = ctypes.c_int
    L.porla_icc_mac_encode_xy_host.argtypes = [u8p, sz, ctypes.c_int, ctypes.c_ulonglong, u8p, u8p]
    L.porla_icc_mac_encode_xy_host.restype = ctypes.c_int
    L.porla_icc_mac_encode_host.argtypes = [u8p, sz, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, u8p]
    L.porla_icc_mac_encode_host.restype = ctypes.c_int
    L.porla_icc_mac_set_matrix_max.argtypes = [sz]; L.porla_icc_mac_set_matrix_max.restype = ctypes.c_int
    L.porla_icc_mix_device.argtypes = [vp, vp, sz, sz, sz, ctypes.c_int, vp, vp]; L.porla_icc_mix_device.restype = ctypes.c_int
    L.porla_icc_mix_host.argtypes = [u8p, u8p, sz, sz, sz, ctypes.c_int, u8p]; L.porla_icc_mix_host.restype = ctypes.c_int
    L.porla_server_mix_device.argtypes = [vp] * 6 + [sz, sz, sz, ctypes.c_int, vp, vp, vp, vp]; L.porla_server_mix_device.restype = ctypes.c_int
    L.porla_icc_mac_mix_pair_device.argtypes = [vp, vp, vp, vp, sz, sz, ctypes.c_int, vp, vp, vp]; L.porla_icc_mac_mix_pair_device.restype = ctypes.c_int
    L.porla_icc_mac_mix_device.argtypes = [vp, vp, sz, sz, ctypes.c_int, vp, vp]; L.porla_icc_mac_mix_device.restype = ctypes.c_int
    L.porla_icc_mac_mix_host.argtypes = [u8p, u8p, sz, sz, ctypes.c_int, u8p]; L.porla_icc_mac_mix_host.restype = ctypes.c_int
    L.porla_audit_combine_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, sz, ctypes.c_int, vp, vp, vp, vp, vp]
    L.porla_audit_combine_device.restype = ctypes.c_int
    L.porla_icc_encode_device.argtypes = [vp, sz, sz, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]
    L.porla_icc_encode_device.restype = ctypes.c_int
    L.porla_icc_encode_host.argtypes = [u8p, sz, sz, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, vp, vp, vp, ctypes.c_int]
    L.porla_icc_encode_host.restype = ctypes.c_int


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load(), name)


lib = _LazyLib()
