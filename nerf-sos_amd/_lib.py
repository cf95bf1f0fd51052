"""ctypes binding of libnerf_sos_hip.so (the C ABI declared in include/nerf_sos_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERF_SOS_HIP_LIB overrides the in-tree library (A/B experiments with alternative kernel builds only)
LIB_PATH = os.environ.get("NERF_SOS_HIP_LIB") or os.path.join(_HERE, "libnerf_sos_hip.so")

_fp = C.c_void_p
_i32, _i64, _f32, _sz = C.c_int32, C.c_int64, C.c_float, C.c_size_t


class MlpTensors(C.Structure):
    """struct nsos_mlp_tensors"""
    _fields_ = [("pts_w", _fp * 8), ("pts_b", _fp * 8), ("alpha_w", _fp), ("alpha_b", _fp),
                ("feature_w", _fp), ("feature_b", _fp), ("views_w", _fp), ("views_b", _fp),
                ("rgb_w", _fp), ("rgb_b", _fp), ("sem0_w", _fp), ("sem0_b", _fp), ("sem2_w", _fp), ("sem2_b", _fp)]


class GenericLinear(C.Structure):
    """struct nsos_generic_linear"""
    _fields_ = [("weight", _fp), ("bias", _fp), ("out_dim", _i32), ("in_dim", _i32)]


GENERIC_MAX_DEPTH, GENERIC_MAX_SEM, GENERIC_LAYOUT_STRIDE = 16, 8, 13


class GenericMlp(C.Structure):
    """struct nsos_generic_mlp"""
    _fields_ = [("depth", _i32), ("width", _i32), ("skip_mask", _i32), ("xyz_freqs", _i32), ("dir_freqs", _i32),
                ("use_viewdirs", _i32), ("use_semantics", _i32), ("sem_dim", _i32), ("sem_with_coord", _i32),
                ("sem_with_geo", _i32), ("sem_layers", _i32),
                ("pts", GenericLinear * GENERIC_MAX_DEPTH), ("alpha", GenericLinear), ("feature", GenericLinear),
                ("views", GenericLinear), ("rgb", GenericLinear), ("output", GenericLinear),
                ("sem", GenericLinear * GENERIC_MAX_SEM), ("geo", GenericLinear * 2)]


class WgradItem(C.Structure):
    """struct nsos_wgrad_item"""
    _fields_ = [("w_off", _i64), ("b_off", _i64), ("g_col", _i32), ("x_col", _i32), ("M", _i32), ("N", _i32), ("ldw", _i32), ("reserved", _i32)]


# name -> (restype, argtypes); must list every symbol the header declares (tests/test_abi.py checks)
SIGNATURES = {
    "nsos_abi_version": (_i32, []),
    "nsos_error_string": (C.c_char_p, [_i32]),
    "nsos_source_hash": (C.c_char_p, []),
    "nsos_mlp_packed_bytes": (_sz, [_i32]),
    "nsos_mlp_pack": (_i32, [C.POINTER(MlpTensors), _i32, _fp, _sz, _fp]),
    "nsos_mlp_generic_packed_bytes": (_sz, [C.POINTER(GenericMlp)]),
    "nsos_mlp_generic_out_channels": (_i32, [C.POINTER(GenericMlp)]),
    "nsos_mlp_generic_pack": (_i32, [C.POINTER(GenericMlp), _fp, _sz, _fp]),
    "nsos_mlp_generic_repack": (_i32, [C.POINTER(GenericMlp), _fp, _sz, _fp]),
    "nsos_mlp_generic_pack_bwd_subset": (_i32, [C.POINTER(GenericMlp), _fp, _sz, C.c_uint32, _i32, _fp]),
    "nsos_mlp_generic_repack_bwd": (_i32, [C.POINTER(GenericMlp), _fp, _sz, _i32, _fp]),
    "nsos_mlp_generic_forward_rays": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp]),
    "nsos_mlp_generic_forward_points": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _i64, _fp, _fp]),
    "nsos_mlp_generic_save_layout": (_i32, [C.POINTER(GenericMlp), C.POINTER(C.c_int32), _i32]),
    "nsos_mlp_generic_forward_rays_save": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_mlp_generic_forward_rays_save_subset": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, C.c_uint32, _fp]),
    "nsos_mlp_generic_bwd_packed_bytes": (_sz, [C.POINTER(GenericMlp), _i32]),
    "nsos_mlp_generic_pack_bwd": (_i32, [C.POINTER(GenericMlp), _fp, _sz, _i32, _fp]),
    "nsos_mlp_generic_input_grads": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _fp, _i64, _fp]),
    "nsos_mlp_generic_input_grads_rays": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_mlp_generic_forward_points_save": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _fp, _i64, _fp, _fp, _fp]),
    "nsos_mlp_generic_input_grads_points": (_i32, [C.POINTER(GenericMlp), _fp, _fp, _fp, _fp, _fp, _fp, _i64, _fp, _fp, _fp, _fp]),
    "nsos_ray_grad_reduce": (_i32, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _f32, _i64, _i32, _i32, _fp, _fp, _fp]),
    "nsos_generate_rays": (_i32, [_i32, _i32, _f32, _f32, _f32, _f32, C.POINTER(C.c_float), _i64, _i64, _fp, _fp, _fp]),
    "nsos_patch_batch": (_i32, [_i32, _i32, _f32, _f32, _f32, _f32, _fp, _i32, _i32, _i32, _fp, _i32, _fp, _i32, C.POINTER(C.c_int32), _fp,
                                _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "nsos_pixel_batch": (_i32, [_i32, _i32, _f32, _f32, _f32, _f32, _fp, _i32, _i32, _i32, _fp, _i32, _fp, _i32, _fp, _i64, _fp, _fp, _fp,
                                _fp, _fp]),
    "nsos_contrastive_loss": (_i32, [_fp, _i32, _i32, _fp, _fp, _fp]),
    "nsos_similarity_negatives": (_i32, [_fp, _i32, _i32, _fp, _fp, _i32, _fp]),
    "nsos_ray_setup": (_i32, [_fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_ray_points": (_i32, [_fp, _fp, _fp, _i64, _i32, _fp, _fp]),
    "nsos_mlp_forward_rays": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp]),
    "nsos_mlp_forward_rays_save": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp]),
    "nsos_mlp_forward_rays_save_all": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_wgrad_workspace_bytes": (_sz, []),
    "nsos_wgrad": (_i32, [_fp, _i32, _fp, _i32, _i64, _i32, _i32, _fp, _i32, _fp, _fp, _sz, _fp]),
    "nsos_wgrad_x3": (_i32, [_fp, _i32, _fp, _i32, _i64, _fp, _i32, _fp, _fp, _sz, _fp]),
    "nsos_wgrad_xh": (_i32, [_fp, _i32, _fp, _i32, _i64, _i32, _i32, _fp, _i32, _fp, _fp, _sz, _fp]),
    "nsos_wgrad_x3_xh": (_i32, [_fp, _i32, _fp, _i32, _i64, _fp, _i32, _fp, _fp, _sz, _fp]),
    "nsos_relu_mask": (_i32, [_fp, _i32, _fp, _i32, _i64, _i32, _fp]),
    "nsos_wgrad_batch": (_i32, [C.POINTER(WgradItem), _i32, _fp, _i32, _fp, _i32, _i64, _fp, _fp, _sz, _fp]),
    "nsos_sem_head_backward": (_i32, [_fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_sem_head_wgrad_workspace_bytes": (_sz, []),
    "nsos_sem_head_wgrad": (_i32, [_fp, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp, _sz, _fp, _i32, _fp]),
    "nsos_sem_head_wgrad_x3": (_i32, [_fp, _fp, _fp, _fp, _fp, _i32, _i64, _i32, _fp, _fp, _fp, _fp, _fp, _sz, _fp, _i32, _fp]),
    "nsos_mlp_packed_bytes_lp": (_sz, [_i32]),
    "nsos_mlp_pack_lp": (_i32, [C.POINTER(MlpTensors), _i32, _i32, _fp, _sz, _fp]),
    "nsos_mlp_pack_lp_heads": (_i32, [C.POINTER(MlpTensors), _i32, _i32, _fp, _sz, _fp]),
    "nsos_mlp_forward_rays_lp": (_i32, [_fp, _i32, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp]),
    "nsos_mlp_forward_rays_save_lp": (_i32, [_fp, _i32, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp]),
    "nsos_mlp_forward_rays_save16_lp": (_i32, [_fp, _i32, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp]),
    "nsos_mlp_save16_layout": (_i32, [_i64]),
    "nsos_mlp_profile_rays": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_mlp_packed_bytes_x3": (_sz, [_i32]),
    "nsos_mlp_pack_x3": (_i32, [C.POINTER(MlpTensors), _i32, _fp, _sz, _fp]),
    "nsos_mlp_forward_rays_x3": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp]),
    "nsos_mlp_forward_rays_save_x3": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp]),
    "nsos_mlp_forward_rays_save_all_x3": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp]),
    "nsos_mlp_forward_rays_save_all16_x3": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp]),
    "nsos_mlp_relu_masks_bytes_x3": (_sz, [_i64]),
    "nsos_mlp_bwd_packed_bytes_x3": (_sz, [_i32]),
    "nsos_mlp_bwd_pack_x3": (_i32, [C.POINTER(MlpTensors), _i32, _fp, _sz, _fp]),
    "nsos_mlp_input_grads_x3": (_i32, [_fp, _i32, _fp, _fp, _fp, _i64, _fp, _fp, _fp]),
    "nsos_mlp_input_grads_x3_a16": (_i32, [_fp, _i32, _fp, _fp, _fp, _i64, _fp, _fp, _fp]),
    "nsos_mlp_profile_rays_x3": (_i32, [_fp, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_mlp_lp_select_kernel": (_i32, [_i32]),
    "nsos_mlp_lp_selected_kernel": (_i32, []),
    "nsos_mlp_x3_select_kernel": (_i32, [_i32]),
    "nsos_mlp_x3_selected_kernel": (_i32, []),
    "nsos_mlp_lp_set_stamp_buffer": (_i32, [_fp]),
    "nsos_mlp_profile_rays_lp": (_i32, [_fp, _i32, _i32, _fp, _fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp]),
    "nsos_mlp_forward_points": (_i32, [_fp, _i32, _fp, _fp, _i64, _fp, _fp]),
    "nsos_composite": (_i32, [_fp, _fp, _fp, _fp, _f32, _i64, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "nsos_composite_importance": (_i32, [_fp, _fp, _fp, _fp, _f32, _i64, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32,
                                         _fp, _fp, _fp, _fp]),
    "nsos_composite_backward": (_i32, [_fp, _fp, _fp, _fp, _f32, _i64, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "nsos_eval_workspace_bytes": (C.c_size_t, []),
    "nsos_eval_postprocess": (_i32, [_fp, _fp, _fp, _i64, _i32, _fp, _fp, _fp, _fp, _fp]),
    "nsos_corr_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "nsos_app_correlation_loss": (_i32, [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                         _f32, _f32, _f32, _f32, _fp, _fp, _fp, _sz, _fp]),
    "nsos_app_correlation_loss_nhwc": (_i32, [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                              _f32, _f32, _f32, _f32, _fp, _fp, _fp, _sz, _fp]),
    "nsos_geo_correlation_loss_pair": (_i32, [_i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32,
                                              _f32, _f32, _f32, _fp, _fp, _fp, _fp, _sz, _fp, _fp, _fp]),
    "nsos_app_correlation_loss_rows": (_i32, [_i32, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                              _f32, _f32, _f32, _f32, _fp, _fp, _fp, _sz, _fp, _fp, _fp]),
    "nsos_corr_exchange_floats": (_i64, [_i32, _i32]),
    "nsos_geo_correlation_loss": (_i32, [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _f32,
                                         _i32, _fp, _fp, _fp, _sz, _fp]),
    "nsos_corr_workspace_slots": (_i32, [_i32, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nsos_geo_correlation_loss_rows": (_i32, [_i32, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32,
                                              _f32, _i32, _fp, _fp, _fp, _sz, _fp, _fp, _fp]),
    "nsos_render_draws": (_i32, [C.c_uint64, C.c_uint64, _i64, _i32, _i32, _fp, _fp, _fp, _fp, _fp]),
    "nsos_render_draws_counted": (_i32, [C.c_uint64, _fp, _i64, _i32, _i32, _fp, _fp, _fp, _fp, _fp]),
    "nsos_importance_sample": (_i32, [_fp, _fp, _fp, _fp, _i64, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp]),
}

ABI_VERSION = 8          # = NSOS_ABI_VERSION of include/nerf_sos_hip.h (an older .so is refused at load)
_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load the HIP library or fail loudly -- there is deliberately no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(a bare `make -C nerf-sos_amd/csrc` also works; it stamps the same source hash).  nerf_sos_amd has no CPU / eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.nsos_abi_version() != ABI_VERSION:
            raise NativeLibraryError(f"ABI version mismatch: library {handle.nsos_abi_version()} vs binding {ABI_VERSION}")
        _lib = handle
    return _lib


def built_source_hash(fresh: bool = False) -> str:
    """The source hash the library on disk was built under (`nsos_source_hash`).  fresh=True asks a child process, so a
    library rebuilt after this process first loaded it is seen (dlopen caches by path)."""
    if not fresh:
        return lib().nsos_source_hash().decode()
    import subprocess
    import sys
    code = ("import ctypes,sys; l=ctypes.CDLL(sys.argv[1]); l.nsos_source_hash.restype=ctypes.c_char_p; "
            "print(l.nsos_source_hash().decode())")
    return subprocess.check_output([sys.executable, "-c", code, LIB_PATH], text=True).strip()


def check(code: int, what: str):
    if code != 0:
        msg = lib().nsos_error_string(code).decode()
        raise RuntimeError(f"{what} failed: [{code}] {msg}")
