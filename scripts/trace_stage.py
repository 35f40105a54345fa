#!/usr/bin/env python3
"""Per-stage timeline of one workgroup of a conv layer (debug build with -DQCNN_TRACE).

Builds quantized-cnn_amd/libqcnn_hip_trace.so when it is missing (needs hipcc), runs AlexNet (synthetic
parameters) on a batch, and prints for the traced workgroup of the chosen layer: the stage period (barrier to
barrier) and, per wave, how long before the barrier opened it had arrived (its slack).  The wave with ~0 slack is the
pole of the stage.   usage: trace_stage.py [layer=0] [block=2000] [batch=1000]
"""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n="": importlib.import_module("quantized-cnn_amd" + ("." + n if n else ""))


def main():
    layer = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    capi, b = pkg("capi"), pkg("build")
    so = os.path.join(capi.PKG, "libqcnn_hip_trace.so")
    if not os.path.exists(so):
        srcs = [os.path.join(b.CSRC, s) for s in b.HIP_SOURCES]
        subprocess.check_call([b.HIPCC] + b.HIP_FLAGS + ["-DQCNN_TRACE", "-shared", "-o", so] + srcs + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"])
    capi.LIB_PATH = so
    topo, synth = pkg("topology"), pkg("synth")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.set_option(capi.OPT_STREAMS, 1)
    eng.load_model(in_chw, layers, params, batch)
    sizes = topo.fmap_sizes(in_chw, layers)
    h, w, c = sizes[layer]
    x = (np.random.default_rng(1).standard_normal((batch, h, w, c)).astype(np.float32)) if layer else None
    buf = (C.c_ulonglong * (16 * 64 * 2 + 16 + 16 * 64))()
    lib = eng.lib
    lib.qcnn_debug_trace_read.argtypes = [C.c_void_p, C.c_int]
    lib.qcnn_debug_trace_read(buf, block)            # select the block
    imgs = synth.make_images(batch, in_chw, seed=2)
    if layer == 0:
        eng.run_layer(0, imgs.transpose(0, 2, 3, 1).copy(), batch)
    else:
        eng.run_layer(layer, x, batch)
    lib.qcnn_debug_trace_read(buf, block)
    t = np.array(buf[: 16 * 64 * 2], dtype=np.float64).reshape(16, 64, 2)
    role = np.array(buf[16 * 64 * 2: 16 * 64 * 2 + 16], dtype=np.int64)
    mid = np.array(buf[16 * 64 * 2 + 16:], dtype=np.float64).reshape(16, 64)
    arrive, leave = t[:, :, 0], t[:, :, 1]
    ok = (arrive[0] > 0).sum()
    print("layer %d block %d batch %d: %d stages traced; roles (100+ = builder): %s" % (layer, block, batch, ok, role.tolist()))
    opens = leave.min(axis=0)                          # the barrier opens when the first wave leaves it
    period = np.diff(opens[:ok])
    print("stage period (cycles): mean %.0f  median %.0f  min %.0f  max %.0f" % (period.mean(), np.median(period), period.min(), period.max()))
    slack = opens[None, :ok] - arrive[:, :ok]          # cycles a wave waited at the barrier
    order = np.argsort(role)
    print("mean wait at the barrier per wave (cycles), builders last:")
    for wv in order:
        print("  wave %2d role %3d : wait %7.0f   busy %7.0f" % (wv, role[wv], slack[wv, 1:].mean(),
                                                                 (arrive[wv, 1:ok] - leave[wv, :ok - 1]).mean()))
    # builders: leave -> [MFMA + stores] -> mid -> [operand loads of the stage after next] -> arrive
    # gather waves: leave -> [offset prefetch issue] -> mid -> [look-ups] -> arrive
    print("split of the busy time at the mid-point (builders: multiply+store | operand loads; gather: prefetch | look-ups):")
    for wv in order:
        a = (mid[wv, 1:ok] - leave[wv, :ok - 1]).mean()
        b2 = (arrive[wv, 1:ok] - mid[wv, 1:ok]).mean()
        print("  wave %2d role %3d : %7.0f | %7.0f" % (wv, role[wv], a, b2))
    print("per stage (first 24): period, slowest gather busy, slowest builder busy")
    for s in range(1, min(ok, 25)):
        busy = arrive[:, s] - leave[:, s - 1]
        gb = busy[role < 100].max()
        bb = busy[role >= 100].max()
        print("  s=%2d period %6.0f gather %6.0f builder %6.0f" % (s, opens[s] - opens[s - 1], gb, bb))


if __name__ == "__main__":
    main()
