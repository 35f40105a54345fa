"""ctypes binding of include/qcnn_hip.h (the C-ABI of libqcnn_hip.so).

Loading the library needs no GPU (symbol checks run on CPU); every compute entry point fails with an
error string when no gfx950 device is present — there is no CPU path behind this module.
"""
from __future__ import annotations

import ctypes as C
import os
import re

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QCNN_HIP_LIB") or os.path.join(PKG, "libqcnn_hip.so")   # QCNN_HIP_LIB: an experimental build of the same library
HEADER_PATH = os.path.join(os.path.dirname(PKG), "include", "qcnn_hip.h")

OPT_LUT_MODE, OPT_KEEP_ALL, OPT_PROFILE, OPT_STREAMS, OPT_SMALL_BATCH, OPT_SPLIT, OPT_HOST_CHUNK, OPT_SLIDE, OPT_DECODE, OPT_SYM, OPT_SYM8, OPT_PACKED_FC, OPT_DIRECT_DEC, OPT_HALF8 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13
LUT_EXACT, LUT_MFMA, LUT_MFMA_F16, LUT_MFMA_F16ACC = 0, 1, 2, 3
SMALL_BATCH_MAX = 3        # QCNN_SMALL_BATCH_MAX (include/qcnn_hip.h)


class QcnnLayerDesc(C.Structure):
    _fields_ = [("type", C.c_int), ("padSiz", C.c_int), ("knlSiz", C.c_int), ("knlCnt", C.c_int),
                ("grpCnt", C.c_int), ("stride", C.c_int), ("nodCnt", C.c_int), ("lrnSiz", C.c_int),
                ("lrnAlp", C.c_float), ("lrnBet", C.c_float), ("lrnIni", C.c_float), ("drpRat", C.c_float)]


def layer_desc(ly: dict) -> QcnnLayerDesc:
    return QcnnLayerDesc(ly["type"], ly.get("pad", 0), ly.get("knl", 0), ly.get("cnt", 0), ly.get("grp", 0),
                         ly.get("stride", 0), ly.get("nod", 0), ly.get("siz", 0), ly.get("alp", 0.0),
                         ly.get("bet", 0.0), ly.get("ini", 0.0), ly.get("rat", 0.0))


def declared_symbols():
    """Entry points declared in include/qcnn_hip.h (parsed from the header text)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qcnn_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load():
    """dlopen libqcnn_hip.so and set prototypes.  Raises a clear error when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no fallback path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i, f32p, u8p, u16p = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p
    lib.qcnn_abi_version.restype = i
    lib.qcnn_device_count.argtypes = [C.POINTER(i)]
    lib.qcnn_last_error.restype = C.c_char_p
    lib.qcnn_last_error.argtypes = [vp]
    lib.qcnn_ctx_create.argtypes = [i, vp, C.POINTER(vp)]
    lib.qcnn_ctx_destroy.argtypes = [vp]
    lib.qcnn_set_option.argtypes = [vp, i, i]
    lib.qcnn_sync.argtypes = [vp]
    lib.qcnn_model_begin.argtypes = [vp, i, C.POINTER(QcnnLayerDesc), i, i, i]
    lib.qcnn_model_set_layer_shape.argtypes = [vp, i, i, i, i]
    lib.qcnn_model_set_layer_dense.argtypes = [vp, i]
    lib.qcnn_model_set_layer_weights.argtypes = [vp, i, f32p, f32p]
    lib.qcnn_group_model_set_layer_dense.argtypes = [vp, i]
    lib.qcnn_group_model_set_layer_weights.argtypes = [vp, i, f32p, f32p]
    lib.qcnn_model_arena_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.qcnn_model_commit.argtypes = [vp, i, vp]
    lib.qcnn_model_set_layer_params.argtypes = [vp, i, f32p, f32p, u8p]
    lib.qcnn_model_set_layer_params_cbn.argtypes = [vp, i, f32p, f32p, u8p, C.c_size_t, i]
    lib.qcnn_model_mark_loaded.argtypes = [vp]
    lib.qcnn_fm_dims.argtypes = [vp, i, C.POINTER(i)]
    lib.qcnn_forward.argtypes = [vp, f32p, i, f32p, u16p]
    lib.qcnn_forward_u8.argtypes = [vp, u8p, i, i, f32p, i, f32p, u16p]
    lib.qcnn_forward_host.argtypes = [vp, f32p, i, f32p, u16p]
    lib.qcnn_forward_host_batches.argtypes = [vp, C.POINTER(vp), C.POINTER(i), i, C.POINTER(vp), C.POINTER(vp)]
    lib.qcnn_host_register.argtypes = [vp, C.c_size_t]
    lib.qcnn_host_unregister.argtypes = [vp]
    lib.qcnn_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.qcnn_host_free.argtypes = [vp]
    lib.qcnn_get_layer_output.argtypes = [vp, i, i, f32p]
    lib.qcnn_get_layer_output_range.argtypes = [vp, i, i, i, f32p]
    lib.qcnn_run_layer.argtypes = [vp, i, f32p, i, f32p]
    lib.qcnn_get_layer_split.argtypes = [vp, i, C.POINTER(i), C.POINTER(i)]
    lib.qcnn_get_layer_segments.argtypes = [vp, i, C.POINTER(i), C.POINTER(i)]
    lib.qcnn_get_layer_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i)]
    lib.qcnn_reset_layer_ms.argtypes = [vp]
    lib.qcnn_get_layer_total_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(i)]
    lib.qcnn_ctx_device.argtypes = [vp]
    lib.qcnn_ctx_stream.argtypes = [vp]
    lib.qcnn_ctx_stream.restype = vp
    lib.qcnn_model_arena_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.qcnn_model_arena_checksum.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    lib.qcnn_plan_conv_query.argtypes = [C.POINTER(i), C.POINTER(i), C.POINTER(C.c_double), C.POINTER(i)]
    # device group
    lib.qcnn_group_create.argtypes = [C.POINTER(i), i, C.POINTER(vp)]
    lib.qcnn_group_destroy.argtypes = [vp]
    lib.qcnn_group_last_error.restype = C.c_char_p
    lib.qcnn_group_last_error.argtypes = [vp]
    lib.qcnn_group_size.argtypes = [vp]
    lib.qcnn_group_ctx.argtypes = [vp, i]
    lib.qcnn_group_ctx.restype = vp
    lib.qcnn_group_shard_bounds.argtypes = [vp, i, i, C.POINTER(i), C.POINTER(i)]
    lib.qcnn_group_set_option.argtypes = [vp, i, i]
    lib.qcnn_group_model_begin.argtypes = [vp, i, C.POINTER(QcnnLayerDesc), i, i, i]
    lib.qcnn_group_model_set_layer_shape.argtypes = [vp, i, i, i, i]
    lib.qcnn_group_model_commit.argtypes = [vp, i]
    lib.qcnn_group_model_set_layer_params.argtypes = [vp, i, f32p, f32p, u8p]
    lib.qcnn_group_model_broadcast.argtypes = [vp, C.POINTER(C.c_float)]
    lib.qcnn_group_arena_checksum.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    lib.qcnn_group_forward_host.argtypes = [vp, f32p, i, f32p, u16p]
    lib.qcnn_group_forward_host_batches.argtypes = [vp, C.POINTER(vp), C.POINTER(i), i, C.POINTER(vp), C.POINTER(vp)]
    lib.qcnn_group_forward.argtypes = [vp, C.POINTER(vp), i, C.POINTER(vp), C.POINTER(vp)]
    lib.qcnn_group_sync.argtypes = [vp]
    _lib = lib
    return lib
