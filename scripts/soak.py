#!/usr/bin/env python3
"""Soak: contexts created, loaded, run and destroyed in a loop; free device memory must return to its level and the
outputs must not change (per mode: the fast path and the layer-for-layer path run different first-layer kernels);
then `repeats` forwards of 1000 images through one context: every one bit-identical to the first.
usage: soak.py [rounds=15] [repeats=60]"""
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
    ref = {}
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
        if r % 2 not in ref:
            ref[r % 2] = (prob.copy(), top5.copy())
        same = np.array_equal(prob, ref[r % 2][0]) and np.array_equal(top5, ref[r % 2][1])
        print("round %2d keep_all %d: outputs identical to the first round of the mode: %s, free %.1f MB%s" %
              (r, r % 2, same, free / 2**20, "" if free0 is None else " (delta %+.1f MB)" % ((free - free0) / 2**20)), flush=True)
        assert same, "outputs changed"
    assert abs(free - free0) < 64 * 2**20, "device memory did not come back"
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    big = synth.make_images(1000, in_chw, seed=3)
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.load_model(in_chw, layers, params, 1000)
    x = torch.from_numpy(big).cuda()
    prob = torch.empty((1000, 1000), dtype=torch.float32, device="cuda")
    first = None
    for r in range(repeats):
        prob.zero_()
        eng.forward_dev(x.data_ptr(), 1000, prob.data_ptr())
        eng.sync()
        h = prob.cpu().numpy()
        if first is None:
            first = h.copy()
            assert np.isfinite(h).all()
        assert np.array_equal(h, first), "forward %d differs from forward 0" % r
    eng.close()
    print("soak OK (%d context rounds, %d repeated 1000-image forwards bit-identical)" % (rounds, repeats))


if __name__ == "__main__":
    main()
