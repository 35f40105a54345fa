"""TEST INFRASTRUCTURE — ctypes bindings of the two CPU checkers.

``COracle``  oracle/libqcnn_oracle.so — the plain-C restatement (qcnn_oracle.c); travels with the repo
             and is (re)built by ``__graft_entry__.build()`` / ``make -C oracle oracle``.
``RefLib``   oracle/_ref/libqcnn_ref.so — the unmodified reference compiled by ``make -C oracle ref``
             where /root/reference exists; the prebuilt .so travels to the GPU box.

Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Nothing else.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libqcnn_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_SO = os.path.join(REF_DIR, "libqcnn_ref.so")
REF_O3_SO = os.path.join(REF_DIR, "libqcnn_ref_o3.so")     # same sources, -O3 -march=znver3: timing comparator only (make ref_o3)
REF_DATA = os.path.join(REF_DIR, "data")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class QoLayer(C.Structure):
    _fields_ = [("type", C.c_int), ("padSiz", C.c_int), ("knlSiz", C.c_int), ("knlCnt", C.c_int),
                ("grpCnt", C.c_int), ("stride", C.c_int), ("nodCnt", C.c_int), ("lrnSiz", C.c_int),
                ("lrnAlp", C.c_float), ("lrnBet", C.c_float), ("lrnIni", C.c_float), ("drpRat", C.c_float)]


def layer_struct(ly: dict) -> QoLayer:
    return QoLayer(ly["type"], ly.get("pad", 0), ly.get("knl", 0), ly.get("cnt", 0), ly.get("grp", 0),
                   ly.get("stride", 0), ly.get("nod", 0), ly.get("siz", 0), ly.get("alp", 0.0),
                   ly.get("bet", 0.0), ly.get("ini", 0.0), ly.get("rat", 0.0))


def build_oracle(force: bool = False) -> str:
    src = os.path.join(HERE, "qcnn_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


class COracle:
    """Whole-network runner + single routines of qcnn_oracle.c."""

    def __init__(self, in_chw, layers):
        self.lib = lib = C.CDLL(build_oracle())
        lib.qo_net_create.restype = C.c_void_p
        lib.qo_net_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(QoLayer)]
        lib.qo_net_destroy.argtypes = [C.c_void_p]
        lib.qo_net_set_params.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _u8p]
        lib.qo_net_set_dense.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        lib.qo_net_fm_dims.argtypes = [C.c_void_p, C.c_int, _i32p]
        lib.qo_net_forward.argtypes = [C.c_void_p, _f32p, C.c_int]
        lib.qo_net_fm.restype = C.POINTER(C.c_float)
        lib.qo_net_fm.argtypes = [C.c_void_p, C.c_int]
        lib.qo_net_run_layer.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_int, _f32p]
        lib.qo_top5.argtypes = [_f32p, C.c_int, _u16p]
        lib.qo_cbn_decode.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
        lib.qo_lut_build.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, _f32p]
        self.in_chw = tuple(in_chw)
        self.layers = layers
        arr = (QoLayer * len(layers))(*[layer_struct(l) for l in layers])
        self.h = lib.qo_net_create(in_chw[0], in_chw[1], in_chw[2], len(layers), arr)
        self.B = 0

    def close(self):
        if self.h:
            self.lib.qo_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        for i, p in params.items():
            m, k, cs = p["ctrd"].shape
            rc = self.lib.qo_net_set_params(self.h, i, np.ascontiguousarray(p["bias"], np.float32),
                                            np.ascontiguousarray(p["ctrd"], np.float32), m, k, cs,
                                            np.ascontiguousarray(p["asmt"], np.uint8))
            if rc:
                raise RuntimeError("qo_net_set_params(%d) -> %d" % (i, rc))

    def set_dense(self, params):
        """Precise path (the reference's Init(false)): {layer: dict(bias, weights)} with conv kernels [Ct][Cg][kh][kw] /
        FC weights [Ct][D] in the convKnl / fcntWei file layout."""
        for i, p in params.items():
            rc = self.lib.qo_net_set_dense(self.h, i, np.ascontiguousarray(p["bias"], np.float32),
                                           np.ascontiguousarray(p["weights"], np.float32).reshape(-1))
            if rc:
                raise RuntimeError("qo_net_set_dense(%d) -> %d" % (i, rc))

    def fm_dims(self, l):
        d = np.zeros(3, np.int32)
        self.lib.qo_net_fm_dims(self.h, l, d)
        return tuple(int(x) for x in d)

    def forward(self, imgs_nchw):
        imgs = np.ascontiguousarray(imgs_nchw, np.float32)
        self.B = imgs.shape[0]
        rc = self.lib.qo_net_forward(self.h, imgs, self.B)
        if rc:
            raise RuntimeError("qo_net_forward -> %d" % rc)

    def fm(self, l):
        h, w, c = self.fm_dims(l)
        n = self.B * h * w * c
        ptr = self.lib.qo_net_fm(self.h, l)
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(self.B, h, w, c).copy()

    def run_layer(self, l, x, batch):
        x = np.ascontiguousarray(x, np.float32)
        h, w, c = self.fm_dims(l + 1)
        out = np.empty((batch, h, w, c), np.float32)
        rc = self.lib.qo_net_run_layer(self.h, l, x, batch, out)
        if rc:
            raise RuntimeError("qo_net_run_layer(%d) -> %d" % (l, rc))
        return out

    def top5(self, prob_row):
        out = np.zeros(5, np.uint16)
        p = np.ascontiguousarray(prob_row, np.float32).reshape(-1)
        self.lib.qo_top5(p, p.size, out)
        return out

    def study_mode(self, lut_f16: bool, acc_f16: bool):
        """Tolerance study switches (process-wide!): always reset to (False, False) afterwards."""
        self.lib.qo_study_mode(int(bool(lut_f16)), int(bool(acc_f16)))


class _Quiet:
    """Silence the reference's printf chatter (fd 1) around a call; bench.py prints ONE JSON line."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        self.null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.null, 1)

    def __exit__(self, *a):
        C.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        os.close(self.null)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class RefLib:
    """The compiled reference (batch 1 only, as the reference is: src/CaffeEva.cc:23)."""

    def __init__(self, so_path: str = REF_SO):
        if not os.path.exists(so_path):
            raise FileNotFoundError(so_path)
        self.lib = lib = C.CDLL(so_path)
        lib.qref_create.restype = C.c_void_p
        lib.qref_destroy.argtypes = [C.c_void_p]
        lib.qref_load_named.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
        lib.qref_load_custom.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, _i32p, _i32p, _f32p]
        if hasattr(lib, "qref_load_custom_prec"):
            lib.qref_load_custom_prec.argtypes = lib.qref_load_custom.argtypes
        lib.qref_layer_cnt.argtypes = [C.c_void_p]
        lib.qref_fm_dims.argtypes = [C.c_void_p, C.c_int, _i32p]
        lib.qref_forward.argtypes = [C.c_void_p, _f32p, _f32p]
        lib.qref_get_fm.argtypes = [C.c_void_p, C.c_int, _f32p]
        lib.qref_lut_elems.argtypes = [C.c_void_p, C.c_int]
        lib.qref_get_lut.argtypes = [C.c_void_p, C.c_int, _f32p]
        lib.qref_run_layer.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        lib.qref_top5.argtypes = [C.c_void_p, _u16p]
        lib.qref_time_forward.restype = C.c_double
        lib.qref_time_forward.argtypes = [C.c_void_p, _f32p, C.c_int, C.POINTER(C.c_double)]
        lib.qref_load_bmp.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, _f32p]
        self.h = lib.qref_create()
        self.L = 0

    def load_bmp(self, mean_path, bmp_path, full=256, crop=227):
        out = np.empty((1, 3, crop, crop), np.float32)
        with _Quiet():
            rc = self.lib.qref_load_bmp(mean_path.encode(), bmp_path.encode(), full, crop, out)
        if rc:
            raise RuntimeError("reference BmpImgIO failed (%d) on %s" % (rc, bmp_path))
        return out

    def load_named(self, model, dir_path, prefix):
        with _Quiet():
            rc = self.lib.qref_load_named(self.h, model.encode(), dir_path.encode(), prefix.encode())
        if rc:
            raise RuntimeError("reference LoadCaffePara failed for %s" % dir_path)
        self.L = self.lib.qref_layer_cnt(self.h)

    def load_custom(self, dir_path, prefix, in_chw, layers, prec=False):
        """prec: the reference's precise path (Init(false): convKnl / fcntWei files, im2col + sgemm)."""
        n = len(layers)
        types = np.array([l["type"] for l in layers], np.int32)
        ip = np.zeros((n, 7), np.int32)
        fp = np.zeros((n, 4), np.float32)
        for i, l in enumerate(layers):
            ip[i] = [l.get("pad", 0), l.get("knl", 0), l.get("cnt", 0), l.get("grp", 0),
                     l.get("stride", 0), l.get("nod", 0), l.get("siz", 0)]
            fp[i] = [l.get("alp", 0.0), l.get("bet", 0.0), l.get("ini", 0.0), l.get("rat", 0.0)]
        with _Quiet():
            fn = self.lib.qref_load_custom_prec if prec else self.lib.qref_load_custom
            rc = fn(self.h, dir_path.encode(), prefix.encode(), in_chw[0], in_chw[1], in_chw[2], n, types, ip, fp)
        if rc:
            raise RuntimeError("reference LoadLayerPara failed for %s" % dir_path)
        self.L = n

    def fm_dims(self, l):
        d = np.zeros(4, np.int32)
        self.lib.qref_fm_dims(self.h, l, d)
        return tuple(int(x) for x in d)

    def forward(self, img_nchw):
        img = np.ascontiguousarray(img_nchw, np.float32).reshape(-1)
        n, h, w, c = self.fm_dims(self.L)
        prob = np.empty(n * h * w * c, np.float32)
        with _Quiet():
            self.lib.qref_forward(self.h, img, prob)
        return prob

    def fm(self, l):
        n, h, w, c = self.fm_dims(l)
        out = np.empty((n, h, w, c), np.float32)
        self.lib.qref_get_fm(self.h, l, out)
        return out

    def lut(self, l):
        n = self.lib.qref_lut_elems(self.h, l)
        out = np.empty(n, np.float32)
        self.lib.qref_get_lut(self.h, l, out)
        return out

    def run_layer(self, l, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        n, h, w, c = self.fm_dims(l + 1)
        out = np.empty((n, h, w, c), np.float32)
        self.lib.qref_run_layer(self.h, l, x, out)
        return out

    def top5(self):
        out = np.zeros(5, np.uint16)
        self.lib.qref_top5(self.h, out)
        return out

    def time_forward(self, imgs_nchw):
        imgs = np.ascontiguousarray(imgs_nchw, np.float32)
        cpu = C.c_double(0.0)
        with _Quiet():
            wall = self.lib.qref_time_forward(self.h, imgs.reshape(-1), imgs.shape[0], C.byref(cpu))
        return wall, cpu.value
