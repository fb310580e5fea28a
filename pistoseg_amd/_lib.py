"""ctypes binding of libpistoseg_hip.so (the C-ABI declared in include/pistoseg_hip.h).

There is NO fallback: if the shared library is missing or a call fails, the op raises.  Build the library
with `python -m pistoseg_amd.build` (or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE our library
# is dlopen'ed, so that both resolve to ONE runtime (streams and device pointers are shared between them);
# loading ours first would pull in /opt/rocm's copy and every launch would fail with hipErrorNoDevice.
import torch  # noqa: F401  (side effect: loads the HIP runtime torch uses)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PISTOSEG_HIP_LIB") or os.path.join(HERE, "libpistoseg_hip.so")  # override: A/B-testing another build

PS_F32, PS_BF16, PS_F16, PS_BF16X3, PS_F16X3 = 0, 1, 2, 3, 4  # PS_*X3: split 16-bit conv formats (blocks of 32 logical channels as [hi(32) | lo(32)]; include/pistoseg_hip.h)
PS_EPI_NONE, PS_EPI_BNRELU, PS_EPI_RELUBWD = 0, 1, 2
PS_MASK_PLAIN, PS_MASK_MUL, PS_MASK_FILL = 0, 1, 2


class PsError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32),
        ("dilation", C.c_int32), ("ldc_x", C.c_int32), ("ldc_y", C.c_int32), ("tiles_per_block", C.c_int32),
        ("gpu_shared", C.c_int32), ("cus_reserved", C.c_int32), ("wgrad_terms", C.c_int32), ("tile_queue", C.c_int32),
    ]


class Epilogue(C.Structure):
    _fields_ = [
        ("add0", C.c_void_p), ("ldc_add0", C.c_int32), ("_pad0", C.c_int32),
        ("out_raw", C.c_void_p), ("ldc_raw", C.c_int32), ("mode", C.c_int32),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("drop", C.c_void_p),
        ("mask_src", C.c_void_p), ("ldc_mask", C.c_int32), ("_pad1", C.c_int32),
        ("add1", C.c_void_p), ("ldc_add1", C.c_int32), ("_pad2", C.c_int32),
        ("out", C.c_void_p), ("ldc_out", C.c_int32), ("_pad3", C.c_int32),
        ("out_hi", C.c_void_p), ("ldc_hi", C.c_int32), ("_pad4", C.c_int32),
        ("sk_ws", C.c_void_p), ("sk_ws_bytes", C.c_int64),
    ]


class DropoutPlan(C.Structure):  # ps_dropout_plan
    _fields_ = [("nseg", C.c_int32), ("_pad", C.c_int32), ("end", C.c_int64 * 8), ("p", C.c_float * 8)]


class WtItem(C.Structure):  # ps_wt_item
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("cout", C.c_int32), ("taps", C.c_int32), ("cin", C.c_int32), ("dst_ld", C.c_int32)]


class TileDst(C.Structure):
    _fields_ = [
        ("canvas", C.c_void_p), ("count", C.c_void_p), ("canvas_h", C.c_int32), ("canvas_w", C.c_int32),
        ("y0", C.c_int32), ("x0", C.c_int32), ("vh", C.c_int32), ("vw", C.c_int32), ("channels_last", C.c_int32), ("_pad", C.c_int32),
    ]


class Tensor4(C.Structure):
    _fields_ = [
        ("ptr", C.c_void_p), ("dtype", C.c_int32), ("n", C.c_int32), ("c", C.c_int32), ("h", C.c_int32),
        ("w", C.c_int32), ("_pad", C.c_int32), ("sn", C.c_int64), ("sc", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64),
    ]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float
_D = C.c_double

# name -> (restype, argtypes); every symbol include/pistoseg_hip.h declares
PROTOTYPES = {
    "ps_version": (C.c_int, []),
    "ps_last_error": (C.c_char_p, []),
    "ps_device_count": (C.c_int, []),
    "ps_conv_supported": (C.c_int, [C.POINTER(ConvGeom)]),
    "ps_conv_variant": (C.c_int, [C.POINTER(ConvGeom), _I]),
    "ps_conv_sk_workspace_bytes": (C.c_int64, [C.POINTER(ConvGeom), _I]),
    "ps_queue_prepare": (C.c_int, [_P]),
    "ps_queue_release": (C.c_int, []),
    "ps_conv_wgrad_variant": (C.c_int, [C.POINTER(ConvGeom)]),
    "ps_conv2d_fwd": (C.c_int, [C.POINTER(ConvGeom), _P, _P, C.POINTER(Epilogue), _P]),
    "ps_conv2d_dgrad": (C.c_int, [C.POINTER(ConvGeom), _P, _P, C.POINTER(Epilogue), _P]),
    "ps_conv2d_wgrad": (C.c_int, [C.POINTER(ConvGeom), _P, _P, _P, _P]),
    "ps_conv2d_wgrad_det_workspace_bytes": (C.c_int64, [C.POINTER(ConvGeom)]),
    "ps_conv2d_wgrad_det": (C.c_int, [C.POINTER(ConvGeom), _P, _P, _P, _P, _L, _P]),
    "ps_weight_transpose": (C.c_int, [_I, _I, _P, _P, _I, _I, _I, _P]),
    "ps_weight_transpose_batched": (C.c_int, [_I, _I, _I, C.POINTER(WtItem), _P]),
    "ps_convert_rows": (C.c_int, [_P, _I, _L, _P, _I, _L, _L, _I, _I, _P]),
    "ps_copy_rows": (C.c_int, [_P, _L, _P, _L, _L, _L, _P]),
    "ps_cast_f32_bf16": (C.c_int, [_P, _P, _L, _P]),
    "ps_cast_f32_lowp": (C.c_int, [_P, _P, _I, _L, _P]),
    "ps_conv1a_fwd": (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ps_fc8_fwd": (C.c_int, [_I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ps_conv1x1_head_workspace_floats": (C.c_int64, [C.POINTER(ConvGeom), _I]),
    "ps_conv1x1_head_fwd": (C.c_int, [C.POINTER(ConvGeom), _P, _P, _P, _P, _P, _I, _P, _L, _P, _P]),
    "ps_fc_head_fwd": (C.c_int, [_I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ps_fc8_bwd": (C.c_int, [_I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "ps_fc8_bwd_workspace_floats": (C.c_int64, [_I, _I, _I, _I]),
    "ps_fc8_bwd_ws": (C.c_int, [_I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, C.c_int64, _P]),
    "ps_bilinear_fwd": (C.c_int, [C.POINTER(Tensor4), C.POINTER(Tensor4), _I, _P]),
    "ps_bilinear_bwd": (C.c_int, [C.POINTER(Tensor4), C.POINTER(Tensor4), _I, _P]),
    "ps_ce_workspace_floats": (C.c_int64, []),
    "ps_softmax_ce": (C.c_int, [_P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P, _P]),
    "ps_dice_workspace_floats": (C.c_int64, []),
    "ps_dice_loss": (C.c_int, [_P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P, _P]),
    "ps_argmax_mask": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ps_confusion_accum": (C.c_int, [_P, _P, _P, _L, _I, _P]),
    "ps_iou_from_confusion": (C.c_int, [_P, _I, _P, _P]),
    "ps_adamw_step": (C.c_int, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P]),
    "ps_sgd_step": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _I, _P]),
    "ps_adamw_step_scaled": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    "ps_adamw_step_guarded": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _F, _F, _P, _F, _P]),
    "ps_sgd_step_scaled": (C.c_int, [_P, _P, _P, _P, _I, _L, _F, _F, _F, _I, _F, _P]),
    "ps_sgd_step_guarded": (C.c_int, [_P, _P, _P, _P, _I, _L, _F, _F, _F, _P, _I, _I, _F, _F, _P]),
    "ps_softmax_scatter_accum": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _P]),
    "ps_canvas_resize_accum": (C.c_int, [_P, _P, _D, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ps_canvas_argmax": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "ps_d4_view": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "ps_scale_inplace": (C.c_int, [_P, _L, _F, _P]),
    "ps_nonfinite_count": (C.c_int, [_P, _L, _P, _P]),
    "ps_dropout2d_masks": (C.c_int, [_P, C.POINTER(DropoutPlan), C.c_uint64, C.c_uint64, _P]),
    "ps_bgemm": (C.c_int, [_I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P]),
    "ps_softmax_rows": (C.c_int, [_P, _L, _I, _P]),
    "ps_rfm_apply": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "ps_affinity_softmax_bwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ps_norm_cam": (C.c_int, [C.POINTER(Tensor4), _P, _L, _L, _L, _P, _I, _P]),
    "ps_loss_workspace_floats": (C.c_int64, []),
    "ps_l1_masked": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _I, _P, _P]),
    "ps_ecr_tensor": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ps_ecr_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P]),
    "ps_tie_workspace_ints": (C.c_int64, [_I, _I, _I]),
    "ps_ecr_bwd_det": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _P, _F, _I, _I, _I, _I, _P]),
    "ps_topk_select": (C.c_int, [_P, _I, _L, _I, _I, _I, _P, _P, _P, _P]),
    "ps_topk_select_workspace_bytes": (C.c_int64, [_I]),
    "ps_topk_select_ws": (C.c_int, [_P, _I, _L, _I, _I, _I, _P, _P, _P, _P, C.c_int64, _P]),
    "ps_sum_scaled": (C.c_int, [_P, _I, _F, _P, _I, _P]),
    "ps_gap": (C.c_int, [_P, _P, _I, _L, _P]),
    "ps_softmargin": (C.c_int, [_P, _P, _P, _P, _I, _F, _I, _I, _P]),
    "ps_gap_bwd": (C.c_int, [_P, _P, _I, _L, _P]),
    "ps_chmax": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ps_minpool_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P]),
    "ps_minpool_bwd_det": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _P, _F, _I, _I, _I, _I, _P]),
}

# symbols of include/pistoseg_hip_debug.h: exported by libpistoseg_hip_debug.so only (-DPS_DEBUG_HOOKS), never by the product library
DEBUG_PROTOTYPES = {
    "ps_debug_hog": (C.c_int, [_I, _I, _I, _P]),
    "ps_debug_set_glds": (None, [C.c_int]),
    "ps_debug_set_3stage": (None, [C.c_int]),
    "ps_debug_set_bn": (None, [C.c_int]),
    "ps_debug_set_bm": (None, [C.c_int]),
    "ps_debug_set_ablate": (None, [C.c_int]),
    "ps_debug_set_pp": (None, [C.c_int]),
    "ps_debug_set_ws": (None, [C.c_int]),
    "ps_debug_set_ws2": (None, [C.c_int]),
    "ps_debug_set_halo": (None, [C.c_int]),
    "ps_debug_set_gemm256": (None, [C.c_int]),
    "ps_debug_set_gemm256_tail": (None, [C.c_int]),
    "ps_debug_set_gemm256_rule": (None, [C.c_int]),
    "ps_debug_set_gemm256_min_tiles": (None, [C.c_int]),
    "ps_debug_set_halo_ring": (None, [C.c_int]),
    "ps_debug_set_halo_tail": (None, [C.c_int]),
    "ps_debug_set_halo_sk": (None, [C.c_int]),
    "ps_debug_set_halo_stagger": (None, [C.c_int]),
    "ps_debug_set_s2split": (None, [C.c_int]),
    "ps_debug_set_wgrad_ws": (None, [C.c_int]),
    "ps_debug_set_wgrad_ws2": (None, [C.c_int]),
    "ps_debug_set_wgrad256": (None, [C.c_int]),
    "ps_debug_set_wgrad_ablate": (None, [C.c_int]),
    "ps_debug_set_wgrad_ovh": (None, [C.c_int]),
    "ps_debug_set_supertile": (None, [C.c_int]),
    "ps_debug_reset": (None, []),
    "ps_debug_set_wgrad_raster": (None, [C.c_int]),
    "ps_debug_set_wgrad_vtab": (None, [C.c_int]),
    "ps_debug_conv_front_s2_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "ps_debug_conv_front_s2": (C.c_int, [C.c_int32] * 4 + [C.c_void_p] * 7 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p]),
}

DEBUG_LIB_PATH = os.environ.get("PISTOSEG_HIP_DEBUG_LIB") or os.path.join(HERE, "libpistoseg_hip_debug.so")  # override: A/B builds
_lib: Optional[C.CDLL] = None
_product: Optional[C.CDLL] = None
_debug: Optional[C.CDLL] = None


def _bind(path: str, protos) -> C.CDLL:
    if not os.path.exists(path):
        raise PsError(
            f"{path} not found: the HIP extension is not built. Run `python -m pistoseg_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = C.CDLL(path)
    for name, (res, args) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise PsError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def load() -> C.CDLL:
    """The library every op launches through: libpistoseg_hip.so (loaded once, every declared symbol bound; raises PsError if
    anything is missing) -- unless a test or tool switched the process to the debug build with `use_debug_library()`."""
    global _lib, _product
    if _lib is None:
        _product = _bind(LIB_PATH, PROTOTYPES)
        _lib = _product
    return _lib


def use_debug_library(on: bool = True) -> C.CDLL:
    """Testing / ablation only: route every op of this process through libpistoseg_hip_debug.so (the same sources built with
    -DPS_DEBUG_HOOKS: the full product ABI plus the `ps_debug_*` switches of include/pistoseg_hip_debug.h), or back."""
    global _lib, _debug
    load()
    if on:
        if _debug is None:
            _debug = _bind(DEBUG_LIB_PATH, {**PROTOTYPES, **DEBUG_PROTOTYPES})
        _lib = _debug
    else:
        _lib = _product
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ps_last_error().decode(errors="replace")
        raise PsError(f"{what} failed (rc={rc}): {msg}")
