"""BASELINE.json configs[3] pinned: full-size VGG-16 (src/CaffePara.cc:121-169) at the batch the VGG-16 number of bench.py
is measured on, through the kernels that measurement uses.

bench.py's `vgg16` block runs 1000 synthetic 224x224 images with the LIBRARY DEFAULTS (decoded first layer, split,
sliding, symmetric and eight-wave symmetric kernels as the planner picks them, fast path, one stream).  Here the same
configuration — same parameters (seed 0), 7 full panels + a ragged one — is checked against the oracle
(src/CaffeEva.cc:760-868, 968-1025, 1261-1296 restated in oracle/qcnn_oracle.c): every feature map the fast path
materialises, the soft-max outputs and the top-5 of images 0, 500 and 999 (first panel, a middle one, the ragged last one), and the
soft-max outputs and top-5 of one mid-panel image of every panel.
Then the kernel FAMILIES' contract: the same batch with the eight-wave, sliding, symmetric and split kernels switched off
(16-wave tile kernels only) must give the same bits on every conv map of those images.
"""
import os

import numpy as np
import pytest

import pyoracle as po
from conftest import pkg, rel_err

pytestmark = pytest.mark.gpu

topo = pkg("topology")
synth = pkg("synth")
capi = pkg("capi")
TOL = 1e-4                                     # north_star: "within 1e-4 relative"
N = int(os.environ.get("QCNN_TEST_VGG_BATCH", "1000"))
IMAGES = (0, N // 2, N - 1)


def _engine(in_chw, layers, params, **opts):
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)       # fast path: ReLU fused into the conv / FC store (what bench.py times)
    eng.set_option(capi.OPT_STREAMS, 1)
    for o, v in opts.items():
        eng.set_option(getattr(capi, o), v)
    eng.load_model(in_chw, layers, params, N)
    return eng


def _maps(eng, L):
    """{feature map index: [len(IMAGES), ...]} of every map the last forward materialised."""
    out = {}
    for l in range(L + 1):
        try:
            out[l] = np.concatenate([eng.layer_output_range(l, i, 1) for i in IMAGES])
        except pkg("engine").QcnnError:
            pass                               # fused away (conv output = the ReLU map behind it) / input read in place
    return out


def test_vgg16_batch_1000_library_defaults_against_oracle_and_tile_kernels():
    in_chw, layers, _, _ = topo.MODELS["VGG16"]
    L = len(layers)
    params = synth.make_params(in_chw, layers, seed=0)          # bench.py's VGG-16 parameters
    imgs = synth.make_images(N, in_chw, seed=54)
    conv = [i for i, l in enumerate(layers) if l["type"] == topo.CONV]

    import torch
    x = torch.from_numpy(imgs).to("cuda:0")                     # device-resident like bench.py's batch: ONE launch per layer over
    prob_d = torch.empty((N, 1000), dtype=torch.float32, device="cuda:0")   # all eight panels (a host batch would go through in
    top5_d = torch.empty((N, 5), dtype=torch.int16, device="cuda:0")        # two-panel chunks, which the planner cuts differently)

    def run(eng):
        eng.forward_dev(x.data_ptr(), N, prob_d.data_ptr(), top5_d.data_ptr())
        eng.sync()
        return prob_d.cpu().numpy(), top5_d.cpu().numpy().view(np.uint16)

    # (a) library defaults at the measured batch size
    eng = _engine(in_chw, layers, params)
    prob, top5 = run(eng)
    codes = {l: eng.layer_split(l)[0] for l in conv}
    # the kernels profiles/r5_vgg16 (and bench.py's vgg16 block) are made of: decoded first layer, the eight-wave sliding
    # form on every layer with >= 128 channels (-6); the 64-channel conv1_2 slides too (16-wave strips -2, or eight-wave -6)
    assert codes[0] == -3, codes
    assert codes[conv[1]] in (-2, -6), codes
    # round 6: the 512-channel layers run half-panel eight-wave workgroups (qcnn_half8.hip: -10 sliding / -9 tile form), the 128- and
    # 256-channel layers keep the full-panel sliding form (-6) — what profiles/r6_vgg16 is made of
    wide = [l for l in conv[2:] if layers[l]["cnt"] == 512]
    assert len(wide) == 6 and all(codes[l] in (-9, -10) for l in wide), codes
    assert all(codes[l] == -6 for l in conv[2:] if l not in wide), codes
    assert np.isfinite(prob).all() and np.abs(prob.sum(axis=1) - 1.0).max() < 1e-4
    got = _maps(eng, L)
    eng.close()
    assert len(got) >= 20, sorted(got)

    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    for j, i in enumerate(IMAGES):
        orc.forward(imgs[i:i + 1])
        for l, fm in got.items():
            e_inf, e_l2 = rel_err(fm[j], orc.fm(l)[0])
            assert e_inf <= TOL and e_l2 <= TOL, "image %d fm[%d]: %g %g" % (i, l, e_inf, e_l2)
        ref = orc.fm(L).reshape(-1)
        e_inf, e_l2 = rel_err(prob[i], ref)
        assert e_inf <= TOL and e_l2 <= TOL, "image %d soft-max: %g %g" % (i, e_inf, e_l2)
        want = orc.top5(ref)
        for a, b in zip(top5[i], want):        # a swap is legitimate only between classes the oracle separates by < TOL
            assert a == b or abs(ref[a] - ref[b]) <= TOL * np.abs(ref).max(), "image %d top-5 %r vs %r" % (i, top5[i], want)
    # one image of EVERY panel (an image in the middle of its panel: lane 77 — the upper half panel of the half-panel kernels), soft-max
    # outputs and top-5 (images 0 / 500 / 999 never sit mid-panel on a boundary the planner chose)
    for pn in range((N + 127) // 128):
        i = min(pn * 128 + 77, N - 1)
        if i in IMAGES:
            continue
        orc.forward(imgs[i:i + 1])
        ref = orc.fm(L).reshape(-1)
        e_inf, e_l2 = rel_err(prob[i], ref)
        assert e_inf <= TOL and e_l2 <= TOL, "panel %d image %d soft-max: %g %g" % (pn, i, e_inf, e_l2)
        want = orc.top5(ref)
        for a, b in zip(top5[i], want):
            assert a == b or abs(ref[a] - ref[b]) <= TOL * np.abs(ref).max(), "image %d top-5 %r vs %r" % (i, top5[i], want)
    orc.close()

    # (b) the families' contract: 16-wave tile kernels only (no eight-wave / sliding / symmetric / split kernels; the first
    # layer stays decoded, the FC layers keep their kernels) — same table entries in the same (kh, kw, m) order per output
    eng = _engine(in_chw, layers, params, OPT_SYM8=0, OPT_HALF8=0, OPT_SLIDE=0, OPT_SYM=0, OPT_SPLIT=0)
    run(eng)
    assert all(eng.layer_split(l)[0] == -1 for l in conv[1:])
    tile = _maps(eng, L)
    eng.close()
    first_fc = [i for i, l in enumerate(layers) if l["type"] == topo.FCNT][0]
    for l in sorted(got):
        if l <= first_fc:                      # everything in front of the FC layers: conv (ReLU-fused) and pool maps
            assert l in tile and np.array_equal(got[l], tile[l]), "fm[%d]: default kernels and tile kernels differ in bits" % l
