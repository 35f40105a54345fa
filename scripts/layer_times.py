#!/usr/bin/env python3
"""Per-layer device time (QCNN_OPT_PROFILE events) of an AlexNet forward at a given batch size.
usage: layer_times.py [batch=128] [steps=20] [streams=1]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("quantized-cnn_amd." + n)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    streams = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    capi, topo, synth = pkg("capi"), pkg("topology"), pkg("synth")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.set_option(capi.OPT_STREAMS, streams)
    eng.load_model(in_chw, layers, params, batch)
    imgs = synth.make_images(batch, in_chw, seed=2)
    import time
    for _ in range(3):
        eng.forward_host(imgs)
    eng.set_option(capi.OPT_PROFILE, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward_host(imgs, want_prob=False)
    wall = (time.perf_counter() - t0) / steps * 1e3
    tot, _, fw = eng.layer_total_ms()
    ms = tot / max(fw, 1)
    names = [topo.TYPE_NAMES[l["type"]] for l in layers]
    print("batch %d streams %d: %.3f ms per forward_host (incl. H2D/D2H), layers sum %.3f ms" % (batch, streams, wall, ms.sum()))
    print("  " + "  ".join("%02d_%s %.3f" % (i, names[i], ms[i]) for i in range(len(layers)) if ms[i] > 0.0005))


if __name__ == "__main__":
    main()
