#!/usr/bin/env python3
"""Soak: contexts created, loaded, run and destroyed in a loop; free device memory must return to its level and the
outputs must not change.  usage: soak.py [rounds=15]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("quantized-cnn_amd." + n)


def main():
    import torch
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    capi, topo, synth = pkg("capi"), pkg("topology"), pkg("synth")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(300, in_chw, seed=2)
    ref = None
    free0 = None
    for r in range(rounds):
        eng = pkg("engine").QcnnEngine(0)
        eng.set_option(capi.OPT_KEEP_ALL, r % 2)
        eng.load_model(in_chw, layers, params, 300)
        prob, top5 = eng.forward_host(imgs)
        p1, t1 = eng.forward_host(imgs[:1])
        eng.close()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info(0)
        if r == 1:
            free0 = free
        if ref is None:
            ref = (prob.copy(), top5.copy())
        same = np.array_equal(prob, ref[0]) and np.array_equal(top5, ref[1])
        print("round %2d keep_all %d: outputs identical to round 0: %s, free %.1f MB%s" %
              (r, r % 2, same, free / 2**20, "" if free0 is None else " (delta %+.1f MB)" % ((free - free0) / 2**20)), flush=True)
        assert same, "outputs changed"
    assert abs(free - free0) < 64 * 2**20, "device memory did not come back"
    print("soak OK")


if __name__ == "__main__":
    main()
