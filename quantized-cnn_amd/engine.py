"""CaffeEva-shaped Python driver over the C-ABI (used by bench.py, the tests and smoke()).

Same life-cycle as the reference's CaffeEva (include/CaffeEva.h:64-85): configure the layer table,
load the per-layer parameters, run forward passes, read per-layer timings — but for a whole batch of
images resident on one MI355X.  All arithmetic happens in libqcnn_hip.so; this file only moves
pointers.  The C++ host mirror (include/CaffeEva.h of this repo) is the same thing for C++ callers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .topology import CONV, FCNT


class QcnnError(RuntimeError):
    pass


class QcnnEngine:
    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = capi.load()
        h = C.c_void_p()
        rc = self.lib.qcnn_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc:
            raise QcnnError(self.lib.qcnn_last_error(None).decode())
        self.h = h
        self.layers = None
        self.L = 0
        self.max_batch = 0

    # -- plumbing -------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise QcnnError(self.lib.qcnn_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.qcnn_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt: int, value: int):
        self._chk(self.lib.qcnn_set_option(self.h, opt, value))

    def sync(self):
        self._chk(self.lib.qcnn_sync(self.h))

    # -- model ----------------------------------------------------------------------------------
    def configure(self, in_chw, layers, shapes):
        """shapes: {layer_idx: (M, K, Cs)} for every conv/FC layer."""
        arr = (capi.QcnnLayerDesc * len(layers))(*[capi.layer_desc(l) for l in layers])
        self._chk(self.lib.qcnn_model_begin(self.h, len(layers), arr, in_chw[0], in_chw[1], in_chw[2]))
        for i, (m, k, cs) in shapes.items():
            self._chk(self.lib.qcnn_model_set_layer_shape(self.h, i, m, k, cs))
        self.layers, self.L, self.in_chw = layers, len(layers), tuple(in_chw)

    def arena_bytes(self) -> int:
        n = C.c_size_t(0)
        self._chk(self.lib.qcnn_model_arena_bytes(self.h, C.byref(n)))
        return n.value

    def commit(self, max_batch: int, arena_ptr: int | None = None):
        self._chk(self.lib.qcnn_model_commit(self.h, max_batch, C.c_void_p(arena_ptr) if arena_ptr else None))
        self.max_batch = max_batch

    def upload(self, params):
        for i, p in params.items():
            bias = np.ascontiguousarray(p["bias"], np.float32)
            ctrd = np.ascontiguousarray(p["ctrd"], np.float32)
            asmt = np.ascontiguousarray(p["asmt"], np.uint8)
            self._chk(self.lib.qcnn_model_set_layer_params(self.h, i, bias.ctypes.data, ctrd.ctypes.data,
                                                           asmt.ctypes.data))

    def upload_cbn(self, params):
        """Same as upload(), but the assignments travel bit-packed (the .cbn payload) and are decoded on the device."""
        from . import fileio
        for i, p in params.items():
            bias = np.ascontiguousarray(p["bias"], np.float32)
            ctrd = np.ascontiguousarray(p["ctrd"], np.float32)
            blocks = fileio.cbn_pack(np.ascontiguousarray(p["asmt"], np.uint8), int(p["bits"]))
            self._chk(self.lib.qcnn_model_set_layer_params_cbn(self.h, i, bias.ctypes.data, ctrd.ctypes.data,
                                                               blocks.ctypes.data, blocks.nbytes, int(p["bits"])))

    def mark_loaded(self):
        self._chk(self.lib.qcnn_model_mark_loaded(self.h))

    def arena_checksum(self):
        """(sum of the arena's 32-bit words, position-weighted sum) as it lies on the device — equal on ranks that hold the same
        parameter bytes (qcnn_model_arena_checksum); blocking."""
        s2 = (C.c_ulonglong * 2)()
        self._chk(self.lib.qcnn_model_arena_checksum(self.h, s2))
        return int(s2[0]), int(s2[1])

    def load_model(self, in_chw, layers, params, max_batch, arena_ptr=None, upload=True):
        shapes = {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()}   # (M, K, Cs)
        for i, ly in enumerate(layers):
            if ly["type"] in (CONV, FCNT) and i not in shapes:
                raise QcnnError("layer %d has no parameters" % i)
        self.configure(in_chw, layers, shapes)
        self.commit(max_batch, arena_ptr)
        if upload:
            self.upload(params)

    def load_dense_model(self, in_chw, layers, dense, max_batch):
        """The reference's precise path (Init(false)): dense = {layer: dict(bias, weights)} in the convKnl / fcntWei file
        layout (synth.make_dense_params)."""
        arr = (capi.QcnnLayerDesc * len(layers))(*[capi.layer_desc(l) for l in layers])
        self._chk(self.lib.qcnn_model_begin(self.h, len(layers), arr, in_chw[0], in_chw[1], in_chw[2]))
        for i in dense:
            self._chk(self.lib.qcnn_model_set_layer_dense(self.h, i))
        self.layers, self.L, self.in_chw = layers, len(layers), tuple(in_chw)
        self.commit(max_batch)
        for i, p in dense.items():
            bias = np.ascontiguousarray(p["bias"], np.float32)
            w = np.ascontiguousarray(p["weights"], np.float32)
            self._chk(self.lib.qcnn_model_set_layer_weights(self.h, i, bias.ctypes.data, w.ctypes.data))

    def fm_dims(self, l):
        d = (C.c_int * 3)()
        self._chk(self.lib.qcnn_fm_dims(self.h, l, d))
        return int(d[0]), int(d[1]), int(d[2])

    # -- forward --------------------------------------------------------------------------------
    def forward_dev(self, in_ptr: int, n: int, prob_ptr: int | None = None, top5_ptr: int | None = None):
        """Asynchronous, device pointers (e.g. torch tensors' data_ptr())."""
        self._chk(self.lib.qcnn_forward(self.h, C.c_void_p(in_ptr), n,
                                        C.c_void_p(prob_ptr) if prob_ptr else None,
                                        C.c_void_p(top5_ptr) if top5_ptr else None))

    def forward_u8_dev(self, in_ptr: int, src_h: int, src_w: int, mean_ptr: int | None, n: int,
                       prob_ptr: int | None = None, top5_ptr: int | None = None):
        """Asynchronous, device pointers: 8-bit planar images [n][C][src_h][src_w], mean [C][src_h][src_w] or None;
        mean subtraction + centre crop happen on the device (BmpImgIO::RmMeanImg/CropImg)."""
        self._chk(self.lib.qcnn_forward_u8(self.h, C.c_void_p(in_ptr), src_h, src_w,
                                           C.c_void_p(mean_ptr) if mean_ptr else None, n,
                                           C.c_void_p(prob_ptr) if prob_ptr else None,
                                           C.c_void_p(top5_ptr) if top5_ptr else None))

    def forward_host(self, imgs_nchw, want_prob=True, want_top5=True):
        imgs = np.ascontiguousarray(imgs_nchw, np.float32)
        n = imgs.shape[0]
        h, w, c = self.fm_dims(self.L)
        prob = np.empty((n, h * w * c), np.float32) if want_prob else None
        top5 = np.empty((n, 5), np.uint16) if want_top5 else None
        self._chk(self.lib.qcnn_forward_host(self.h, imgs.ctypes.data, n,
                                             prob.ctypes.data if want_prob else None,
                                             top5.ctypes.data if want_top5 else None))
        return prob, top5

    def forward_host_batches(self, batches, want_prob=True, want_top5=True):
        """Blocking: a list of [n_b, C, H, W] float32 arrays, one batch after the other with the upload of batch b + 1
        overlapped with the layers of batch b (qcnn_forward_host_batches).  Returns ([prob_b], [top5_b])."""
        return _host_batches(self.lib.qcnn_forward_host_batches, self, batches, self.fm_dims(self.L), want_prob, want_top5)

    def layer_output(self, l: int, n: int):
        h, w, c = self.fm_dims(l)
        out = np.empty((n, h, w, c), np.float32)
        self._chk(self.lib.qcnn_get_layer_output(self.h, l, n, out.ctypes.data))
        return out

    def layer_output_range(self, l: int, first: int, n: int):
        h, w, c = self.fm_dims(l)
        out = np.empty((n, h, w, c), np.float32)
        self._chk(self.lib.qcnn_get_layer_output_range(self.h, l, first, n, out.ctypes.data))
        return out

    def run_layer(self, l: int, x, n: int):
        x = np.ascontiguousarray(x, np.float32)
        h, w, c = self.fm_dims(l + 1)
        out = np.empty((n, h, w, c), np.float32)
        self._chk(self.lib.qcnn_run_layer(self.h, l, x.ctypes.data, n, out.ctypes.data))
        return out

    def layer_split(self, l: int):
        """(tiles run whole, slices per split tile) of the last launch of conv layer l (QCNN_OPT_SPLIT); slices = 1: no split."""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self.lib.qcnn_get_layer_split(self.h, l, C.byref(a), C.byref(b)))
        return a.value, b.value

    def layer_segments(self, l: int):
        """Row-segment boundaries of the sliding kernel's last launch of conv layer l ([] = the tile kernel ran)."""
        seg = (C.c_int * 9)()
        n = C.c_int(0)
        self._chk(self.lib.qcnn_get_layer_segments(self.h, l, seg, C.byref(n)))
        return [int(seg[i]) for i in range(n.value + 1)] if n.value > 0 else []

    # -- timing ---------------------------------------------------------------------------------
    def layer_ms(self):
        ms = (C.c_float * self.L)()
        cnt = C.c_int(0)
        self._chk(self.lib.qcnn_get_layer_ms(self.h, ms, C.byref(cnt)))
        return np.array(ms[:], np.float64), cnt.value

    def layer_total_ms(self):
        """(ms summed over every recorded launch, launches) per layer, forwards recorded."""
        tot = (C.c_double * self.L)()
        cnt = (C.c_longlong * self.L)()
        fw = C.c_int(0)
        self._chk(self.lib.qcnn_get_layer_total_ms(self.h, tot, cnt, C.byref(fw)))
        return np.array(tot[:], np.float64), np.array(cnt[:], np.int64), fw.value

    def reset_layer_ms(self):
        self._chk(self.lib.qcnn_reset_layer_ms(self.h))


def _host_batches(fn, obj, batches, out_hwc, want_prob, want_top5):
    imgs = [np.ascontiguousarray(b, np.float32) for b in batches]
    nb = len(imgs)
    classes = out_hwc[0] * out_hwc[1] * out_hwc[2]
    prob = [np.empty((b.shape[0], classes), np.float32) for b in imgs] if want_prob else None
    top5 = [np.empty((b.shape[0], 5), np.uint16) for b in imgs] if want_top5 else None
    vp = C.c_void_p
    a_in = (vp * nb)(*[b.ctypes.data for b in imgs])
    a_n = (C.c_int * nb)(*[b.shape[0] for b in imgs])
    a_p = (vp * nb)(*[p.ctypes.data for p in prob]) if want_prob else None
    a_t = (vp * nb)(*[t.ctypes.data for t in top5]) if want_top5 else None
    obj._chk(fn(obj.h, a_in, a_n, nb, a_p, a_t))
    return prob, top5


def host_register(arr) -> None:
    """Pin a numpy array's storage (hipHostRegister): uploads from it become asynchronous DMA transfers."""
    lib = capi.load()
    if lib.qcnn_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes):
        raise QcnnError(lib.qcnn_last_error(None).decode())


def host_unregister(arr) -> None:
    lib = capi.load()
    if lib.qcnn_host_unregister(C.c_void_p(arr.ctypes.data)):
        raise QcnnError(lib.qcnn_last_error(None).decode())


class QcnnDeviceGroup:
    """One batch sharded over several GPUs of this process (qcnn_group_* of include/qcnn_hip.h): contiguous image
    blocks, parameters uploaded to rank 0 and broadcast to the others with RCCL, one host thread per GPU."""

    def __init__(self, devices=None):
        self.lib = capi.load()
        h = C.c_void_p()
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            rc = self.lib.qcnn_group_create(arr, len(devices), C.byref(h))
        else:
            rc = self.lib.qcnn_group_create(None, 0, C.byref(h))
        if rc:
            raise QcnnError(self.lib.qcnn_group_last_error(None).decode())
        self.h = h
        self.size = self.lib.qcnn_group_size(h)
        self.L = 0
        self.classes = 0
        self.broadcast_ms = None

    def _chk(self, rc):
        if rc:
            raise QcnnError(self.lib.qcnn_group_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.qcnn_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, value):
        self._chk(self.lib.qcnn_group_set_option(self.h, opt, value))

    def shard_bounds(self, n, rank):
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self.lib.qcnn_group_shard_bounds(self.h, n, rank, C.byref(a), C.byref(b)))
        return a.value, a.value + b.value

    def load_model(self, in_chw, layers, params, max_batch):
        shapes = {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()}
        arr = (capi.QcnnLayerDesc * len(layers))(*[capi.layer_desc(l) for l in layers])
        self._chk(self.lib.qcnn_group_model_begin(self.h, len(layers), arr, in_chw[0], in_chw[1], in_chw[2]))
        for i, (m, k, cs) in shapes.items():
            self._chk(self.lib.qcnn_group_model_set_layer_shape(self.h, i, m, k, cs))
        self._chk(self.lib.qcnn_group_model_commit(self.h, max_batch))
        for i, p in params.items():
            bias = np.ascontiguousarray(p["bias"], np.float32)
            ctrd = np.ascontiguousarray(p["ctrd"], np.float32)
            asmt = np.ascontiguousarray(p["asmt"], np.uint8)
            self._chk(self.lib.qcnn_group_model_set_layer_params(self.h, i, bias.ctypes.data, ctrd.ctypes.data,
                                                                 asmt.ctypes.data))
        self.broadcast()
        self.L = len(layers)
        d = (C.c_int * 3)()
        self.lib.qcnn_fm_dims(self.lib.qcnn_group_ctx(self.h, 0), self.L, d)
        self.classes = int(d[0]) * int(d[1]) * int(d[2])

    def upload(self, params):
        """Re-upload layers' parameters to rank 0 ({layer: dict(bias, ctrd, asmt)}); broadcast() must follow."""
        for i, p in params.items():
            bias = np.ascontiguousarray(p["bias"], np.float32)
            ctrd = np.ascontiguousarray(p["ctrd"], np.float32)
            asmt = np.ascontiguousarray(p["asmt"], np.uint8)
            self._chk(self.lib.qcnn_group_model_set_layer_params(self.h, i, bias.ctypes.data, ctrd.ctypes.data, asmt.ctypes.data))

    def broadcast(self):
        """Rank 0's arena to every rank (RCCL), verified by a per-rank device checksum (qcnn_group_model_broadcast)."""
        ms = C.c_float(0.0)
        self._chk(self.lib.qcnn_group_model_broadcast(self.h, C.byref(ms)))
        self.broadcast_ms = ms.value

    def arena_checksum(self):
        s2 = (C.c_ulonglong * 2)()
        self._chk(self.lib.qcnn_group_arena_checksum(self.h, s2))
        return int(s2[0]), int(s2[1])

    def forward_host(self, imgs_nchw):
        imgs = np.ascontiguousarray(imgs_nchw, np.float32)
        n = imgs.shape[0]
        prob = np.empty((n, self.classes), np.float32)
        top5 = np.empty((n, 5), np.uint16)
        self._chk(self.lib.qcnn_group_forward_host(self.h, imgs.ctypes.data, n, prob.ctypes.data, top5.ctypes.data))
        return prob, top5

    def forward_dev(self, in_ptrs, n, prob_ptrs=None, top5_ptrs=None):
        """Asynchronous, device-resident (qcnn_group_forward): per-rank device pointers to each rank's block; sync() waits."""
        vp = C.c_void_p
        arr = lambda ptrs: (vp * self.size)(*[vp(p) if p else None for p in ptrs]) if ptrs is not None else None
        self._chk(self.lib.qcnn_group_forward(self.h, arr(in_ptrs), n, arr(prob_ptrs), arr(top5_ptrs)))

    def sync(self):
        self._chk(self.lib.qcnn_group_sync(self.h))

    def forward_host_batches(self, batches, want_prob=True, want_top5=True):
        d = (C.c_int * 3)()
        self.lib.qcnn_fm_dims(self.lib.qcnn_group_ctx(self.h, 0), self.L, d)
        return _host_batches(self.lib.qcnn_group_forward_host_batches, self, batches, (int(d[0]), int(d[1]), int(d[2])),
                             want_prob, want_top5)
