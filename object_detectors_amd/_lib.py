"""ctypes loader for liblmi_hip.so (the C ABI of include/lmi_hip.h).  No fallback: a missing or
stale library raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must come first: the library has to bind to the HIP runtime torch ships and uses

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblmi_hip.so")

LMI_F32, LMI_F16, LMI_U8 = 0, 1, 2
LMI_MAX_DET_CAP = 1024
LMI_NMS_STATIC_BYTES = 4 << 20


class LmiError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"liblmi_hip: {message} (status {status})")
        self.status = status


class lmi_frame(C.Structure):
    _fields_ = [("src", C.c_uint64), ("h", C.c_int32), ("w", C.c_int32), ("row_stride", C.c_int32),
                ("new_w", C.c_int32), ("new_h", C.c_int32), ("pad_l", C.c_int32), ("pad_t", C.c_int32),
                ("reserved", C.c_int32)]


class lmi_finalize_args(C.Structure):
    _fields_ = [("model_h", C.c_int32), ("model_w", C.c_int32), ("uniform_h0", C.c_int32), ("uniform_w0", C.c_int32),
                ("orig_hw", C.c_void_p), ("pad_round", C.c_int32), ("reserved", C.c_int32), ("thres", C.c_void_p),
                ("uniform_thres", C.c_double), ("keep", C.c_void_p), ("keep_count", C.c_void_p), ("mask_slot", C.c_void_p),
                ("slot_map", C.c_void_p), ("slot_base", C.c_void_p),
                ("kpt_col0", C.c_int32), ("kpt_n", C.c_int32), ("kpt_ndim", C.c_int32), ("reserved2", C.c_int32)]


class lmi_mask_image(C.Structure):
    _fields_ = [("h0", C.c_int32), ("w0", C.c_int32), ("n_masks", C.c_int32), ("reserved", C.c_int32), ("out_offset", C.c_int64)]


LMI_MASK_LOGITS_MFMA, LMI_MASK_UPSAMPLE_SINGLE, LMI_MASK_PROTO_F16 = 1, 2, 4
LMI_SRC_GRAY = 0x100
LMI_LB_TABLES_ONLY, LMI_LB_TABLES_READY = 0x200, 0x400
LMI_NMS_SWEEP_ONLY = 0x100


class lmi_point_op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("x", C.c_float), ("y", C.c_float)]


LMI_POINT_DIV, LMI_POINT_SUB, LMI_POINT_FLIP_X, LMI_POINT_FLIP_Y, LMI_POINT_FLIP_XY = 1, 2, 3, 4, 5
LMI_MAX_POINT_OPS = 16


# name -> (restype, argtypes); must list every symbol include/lmi_hip.h declares
SIGNATURES = {
    "lmi_version": (C.c_int, []),
    "lmi_status_string": (C.c_char_p, [C.c_int]),
    "lmi_last_error": (C.c_char_p, []),
    "lmi_device_count": (C.c_int, []),
    "lmi_preprocess_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "lmi_letterbox_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "lmi_letterbox_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "lmi_letterbox_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "lmi_letterbox_u8_uniform": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_int64, C.c_void_p]),
    "lmi_nms_workspace_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "lmi_nms_v8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int,
                             C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "lmi_nms_v8_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int,
                                C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_void_p]),
    "lmi_nms_decode_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int64,
                                    C.c_void_p]),
    "lmi_nms_suppress_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p]),
    "lmi_nms_v8_static": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int,
                                    C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lmi_nms_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "lmi_nms_suppress": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "lmi_finalize_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lmi_nms_v8_rotated": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int,
                                     C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "lmi_finalize_obb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lmi_scale_coords": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lmi_mask_logits_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                     C.c_void_p]),
    "lmi_mask_upsample_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_int, C.c_void_p]),
    "lmi_process_mask_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "lmi_mask_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "lmi_mask_logits": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "lmi_mask_upsample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "lmi_process_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p]),
    "lmi_paste_masks": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_void_p, C.c_void_p]),
    "lmi_pack_mask_boxes": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lmi_trace_contours_host": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                            C.c_void_p]),
    "lmi_revert_points_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "lmi_revert_masks_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises if it is missing: build it with
    `python -c "import __graft_entry__ as g; g.build()"` or `make -C object_detectors_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP extension is not built (make -C object_detectors_amd/csrc); "
                          "there is no CPU fallback for this path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        lib = load()
        msg = lib.lmi_last_error().decode() or lib.lmi_status_string(status).decode()
        if status == -1:
            raise ValueError(f"liblmi_hip: {msg}")
        raise LmiError(status, msg)
