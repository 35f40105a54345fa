#!/usr/bin/env python3
"""Per-layer device time (QCNN_OPT_PROFILE events) of a forward at a given batch size.
usage: [QCNN_MODEL=AlexNet|VGG16] [QCNN_SPLIT|QCNN_SLIDE|QCNN_SYM|QCNN_SYM8|QCNN_HALF8|QCNN_DIRECT_DEC|QCNN_DECODE|QCNN_LUT=..] layer_times.py [batch=128] [steps=20] [streams=1]"""
import importlib, os, sys
import numpy as np
import torch   # before libqcnn_hip.so: both must bind to the HIP runtime torch ships
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("quantized-cnn_amd." + n)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    streams = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    capi, topo, synth = pkg("capi"), pkg("topology"), pkg("synth")
    in_chw, layers, _, _ = topo.MODELS[os.environ.get("QCNN_MODEL", "AlexNet")]
    params = synth.make_params(in_chw, layers, seed=0)
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.set_option(capi.OPT_STREAMS, streams)
    split = int(os.environ.get("QCNN_SPLIT", "1"))
    eng.set_option(capi.OPT_SPLIT, split)
    eng.set_option(capi.OPT_SLIDE, int(os.environ.get("QCNN_SLIDE", "1")))
    eng.set_option(capi.OPT_SYM, int(os.environ.get("QCNN_SYM", "1")))
    eng.set_option(capi.OPT_SYM8, int(os.environ.get("QCNN_SYM8", "1")))
    eng.set_option(capi.OPT_HALF8, int(os.environ.get("QCNN_HALF8", "1")))
    eng.set_option(capi.OPT_DIRECT_DEC, int(os.environ.get("QCNN_DIRECT_DEC", "1")))
    eng.set_option(capi.OPT_DECODE, int(os.environ.get("QCNN_DECODE", "1")))
    eng.set_option(capi.OPT_LUT_MODE, int(os.environ.get("QCNN_LUT", "1")))
    eng.load_model(in_chw, layers, params, batch)
    imgs = synth.make_images(batch, in_chw, seed=2)
    import time
    x = torch.from_numpy(imgs).cuda()
    top5 = torch.empty((batch, 5), dtype=torch.int16, device="cuda")
    import time
    for _ in range(3):
        eng.forward_dev(x.data_ptr(), batch, None, top5.data_ptr())
    eng.sync()
    eng.set_option(capi.OPT_PROFILE, 1)
    for _ in range(steps):
        eng.forward_dev(x.data_ptr(), batch, None, top5.data_ptr())
    eng.sync()
    tot, _, fw = eng.layer_total_ms()
    ms = tot / max(fw, 1)
    cuts = " ".join("%d:%dx%d" % ((l,) + eng.layer_split(l)) for l in [i for i, ly in enumerate(layers) if ly["type"] == topo.CONV])
    t0 = time.perf_counter()
    eng.forward_host(imgs, want_prob=False)
    wall_first = (time.perf_counter() - t0) * 1e3     # the first call allocates the staging buffers and plans the chunk launches
    t0 = time.perf_counter()
    eng.forward_host(imgs, want_prob=False)
    wall = (time.perf_counter() - t0) * 1e3
    names = [topo.TYPE_NAMES[l["type"]] for l in layers]
    eng.set_option(capi.OPT_PROFILE, 0)
    for _ in range(3):
        eng.forward_dev(x.data_ptr(), batch, None, top5.data_ptr())
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward_dev(x.data_ptr(), batch, None, top5.data_ptr())
    eng.sync()
    dev = (time.perf_counter() - t0) / steps * 1e3
    print("batch %d streams %d split %d: resident %.3f ms (%.0f img/s), forward_host %.3f ms (first call %.1f), layers sum %.3f ms, cuts %s"
          % (batch, streams, split, dev, batch / dev * 1e3, wall, wall_first, ms.sum(), cuts))
    print("  " + "  ".join("%02d_%s %.3f" % (i, names[i], ms[i]) for i in range(len(layers)) if ms[i] > 0.0005))


if __name__ == "__main__":
    main()
