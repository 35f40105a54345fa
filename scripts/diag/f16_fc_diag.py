#!/usr/bin/env python3
"""Diagnostic (round 5): where do the fp16-storage FC kernel (k_fc_sym8<true>) and the rounded-f32 FC kernel differ?  fc7 of AlexNet in
isolation on three kinds of input — ordinary magnitudes, magnitudes whose table entries are fp16 subnormals, exact zeros — against each
other and against the oracle's study mode (entries rounded to fp16, nearest even, subnormals kept)."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
pkg = lambda n: importlib.import_module("quantized-cnn_amd." + n)
capi, topo, synth = pkg("capi"), pkg("topology"), pkg("synth")
in_chw, layers, _, _ = topo.MODELS["AlexNet"]
params = synth.make_params(in_chw, layers, seed=7)
L = 18                                   # fc7
n = 5
def engine(sym8):
    e = pkg("engine").QcnnEngine(0)
    e.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA_F16); e.set_option(capi.OPT_KEEP_ALL, 1); e.set_option(capi.OPT_SPLIT, 0)
    e.set_option(capi.OPT_DECODE, 0); e.set_option(capi.OPT_SYM8, sym8)
    e.load_model(in_chw, layers, params, 8)
    return e
real, emu = engine(1), engine(0)
orc = po.COracle(in_chw, layers); orc.set_params(params)
rng = np.random.default_rng(5)
for name, x in (("ordinary |x| ~ 1", rng.standard_normal((n, 1, 1, 4096)).astype(np.float32)),
                ("relu'd, many zeros", np.maximum(rng.standard_normal((n, 1, 1, 4096)), 0).astype(np.float32) * 3),
                ("tiny |x| ~ 1e-4 (entries fp16-subnormal)", (rng.standard_normal((n, 1, 1, 4096)) * 1e-4).astype(np.float32)),
                ("tiny |x| ~ 1e-6 (entries below the fp16 quantum)", (rng.standard_normal((n, 1, 1, 4096)) * 1e-6).astype(np.float32))):
    a = real.run_layer(L, x, n); b = emu.run_layer(L, x, n)
    orc.study_mode(True, False); o16 = orc.run_layer(L, x, n); orc.study_mode(False, False); o32 = orc.run_layer(L, x, n)
    bias = params[L]["bias"].reshape(1, 1, 1, -1)
    sc = np.abs(o32 - bias).max()
    print("%-50s split real %s emu %s | scale (sum part) %.3g | real-emu %.2e  real-orc16 %.2e  emu-orc16 %.2e  orc16-orc32 %.2e (all / scale)"
          % (name, real.layer_split(L), emu.layer_split(L), sc, np.abs(a - b).max() / sc, np.abs(a - o16).max() / sc, np.abs(b - o16).max() / sc,
             np.abs(o16 - o32).max() / sc), flush=True)
