#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — one process of bench.py's `cpu_baseline` all-cores / -O3 legs.

The reference is single-threaded and not re-entrant (function-local statics, src/CaffeEva.cc:415): "all cores" therefore means
one PROCESS of the compiled reference per core, each classifying its own images one at a time (the reference's regime,
kDataCntInBatch = 1, src/CaffeEva.cc:23).  This worker loads the parameter files of <param_dir>, waits for the common start
time, times `n` forward passes with the reference's own stop-watch and prints one JSON line.

usage: ref_worker.py <so_path> <param_dir> <prefix> <n_images> <start_epoch_seconds>
Executed only by bench.py's cpu_baseline leg; nothing in the product path imports it."""
import importlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    so, pdir, prefix, n, start = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5])
    import numpy as np
    import pyoracle as po
    topo = importlib.import_module("quantized-cnn_amd.topology")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    ref = po.RefLib(so)
    ref.load_custom(pdir, prefix, in_chw, layers)
    rng = np.random.default_rng(os.getpid())
    imgs = (rng.integers(0, 256, size=(n,) + tuple(in_chw)).astype(np.float32) - 117.0)
    ref.time_forward(imgs[:1])                       # page in
    while time.time() < start:
        time.sleep(0.005)
    t0 = time.time()
    wall, cpu = ref.time_forward(imgs)
    print(json.dumps(dict(n=n, wall=wall, cpu=cpu, t0=t0, t1=time.time())), flush=True)


if __name__ == "__main__":
    main()
