#!/usr/bin/env python3
"""Per-stage timeline of one workgroup of k_conv_sym8 (debug build of qcnn_sym8.hip with -DS8_TRACE: scripts/build_variant.sh
_s8trace qcnn_sym8.hip -DS8_TRACE).  For every wave: mean cycles of the build phase (matrix instructions + LDS stores + issue
of the next operand loads), of the look-up phase, and of the wait at the stage barrier; the stage period.
usage: trace_sym8.py [layer=8] [block=300] [batch=1000] [sym8 mode=2]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n="": importlib.import_module("quantized-cnn_amd" + ("." + n if n else ""))


def main():
    layer = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    capi = pkg("capi")
    capi.LIB_PATH = os.path.join(capi.PKG, "libqcnn_hip_s8trace.so")
    topo, synth = pkg("topology"), pkg("synth")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.set_option(capi.OPT_STREAMS, 1)
    eng.set_option(capi.OPT_SYM8, mode)
    eng.load_model(in_chw, layers, params, batch)
    h, w, c = topo.fmap_sizes(in_chw, layers)[layer]
    x = np.random.default_rng(1).standard_normal((batch, h, w, c)).astype(np.float32)
    buf = (C.c_ulonglong * 64)()
    lib = eng.lib
    lib.qcnn_debug_trace8_read.argtypes = [C.c_void_p, C.c_int]
    lib.qcnn_debug_trace8_read(buf, block)
    eng.run_layer(layer, x, batch)
    assert eng.layer_split(layer)[0] == -5, eng.layer_split(layer)
    lib.qcnn_debug_trace8_read(buf, block)
    t = np.array(buf[:], dtype=np.float64).reshape(8, 8)
    tot = t[:, 1:5].sum(axis=1)
    print("layer %d block %d batch %d mode %d: cycles per wave over the workgroup's whole stage loop" % (layer, block, batch, mode))
    print("           multiply + store | operand loads issued | look-ups | barrier |   total")
    for wv in range(8):
        print("  wave %d : %8.0f (%4.1f %%) | %8.0f (%4.1f %%) | %8.0f (%4.1f %%) | %8.0f (%4.1f %%) | %8.0f" % (
            wv, t[wv, 4], 100 * t[wv, 4] / tot[wv], t[wv, 1], 100 * t[wv, 1] / tot[wv], t[wv, 2], 100 * t[wv, 2] / tot[wv],
            t[wv, 3], 100 * t[wv, 3] / tot[wv], tot[wv]))


if __name__ == "__main__":
    main()
