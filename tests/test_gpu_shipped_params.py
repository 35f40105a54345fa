"""GPU parity on the SHIPPED AlexNet parameters at the headline batch size, through the kernels the headline runs.

north_star: "outputs match the reference CPU CaffeEva layer-for-layer within 1e-4 relative on the same 227x227
inputs".  The ten shipped Bmp.Files/*.BMP (decoded by the reference's own BmpImgIO) are tiled to 1000 images = 7 full
panels + a ragged one of 104, and every feature map the configuration materialises is compared with what the COMPILED
reference produced for that image (tests/golden/alexnet_real10_ref.npz, oracle/make_golden.py: samples + l2 norm; and every
element of every map against the oracle, which is pinned bit for bit to the compiled reference) — for images of the
first panel, of a middle one and of the ragged last one.  Configurations: the library defaults (decoded conv1 / fc8,
split, sliding, symmetric and eight-wave symmetric kernels as the planner picks them), layer-for-layer and fast path (fused ReLU, fused
LRN + pool, one stream = what bench.py times), every eligible layer forced through the symmetric / sliding kernels, and
every layer through tables (QCNN_OPT_DECODE = 0).  The fc6 assignment file is missing from the reference mount
(SURVEY.md §0 fact 3): fixture 1 (SURVEY's recipe) leaves fc7 / fc8 degenerate, so the tail fm[16..23], the soft-max
outputs and the top-5 are checked with fixture 2 (synth.fc6_fixture), whose tail differs from image to image.
"""
import os

import numpy as np
import pytest

import pyoracle as po
from conftest import pkg, real_bmp_images

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isdir(po.REF_DATA), reason="shipped parameters not staged (oracle/_ref/data)")]

topo = pkg("topology")
synth = pkg("synth")
capi = pkg("capi")
TOL = 1e-4
SAMPLE_STRIDE = 97
N = 1000
BLOCKS = ((0, 10), (500, 10), (990, 10))       # (first image, count): first panel, a middle one, the ragged last one


def _engine(params, keep_all, streams=None, host_chunk=None, sym=None, slide=None, decode=None, split=None, sym8=None, half8=None):
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA)
    eng.set_option(capi.OPT_KEEP_ALL, keep_all)
    for opt, v in ((capi.OPT_STREAMS, streams), (capi.OPT_HOST_CHUNK, host_chunk), (capi.OPT_SYM, sym),
                   (capi.OPT_SLIDE, slide), (capi.OPT_DECODE, decode), (capi.OPT_SPLIT, split), (capi.OPT_SYM8, sym8), (capi.OPT_HALF8, half8)):
        if v is not None:
            eng.set_option(opt, v)
    eng.load_model(in_chw, layers, params, N)
    return eng


def _check_maps(eng, z, fx, maps, tag):
    """Every map of `maps` the run materialised, for the images of BLOCKS, against the reference's samples and l2 norm."""
    checked = 0
    for l in maps:
        for first, cnt in BLOCKS:
            try:
                fm = eng.layer_output_range(l, first, cnt).reshape(cnt, -1)
            except pkg("engine").QcnnError:
                checked -= 1                             # fused away / not materialised in this configuration
                break
            for j in range(cnt):
                i = (first + j) % 10
                fp = z["fp%d_%02d" % (fx, l)][i]
                scale = max(abs(fp[3]), abs(fp[4]), 1e-30)
                err = np.abs(fm[j, ::SAMPLE_STRIDE].astype(np.float64) - z["smp%d_%02d" % (fx, l)][i]).max() / scale
                assert err <= TOL, "%s: fixture %d fm[%d] image %d: %g" % (tag, fx, l, first + j, err)
                l2 = np.sqrt((fm[j].astype(np.float64) ** 2).sum())
                assert abs(l2 - fp[2]) <= TOL * max(fp[2], 1e-30), "%s: fixture %d fm[%d] image %d l2" % (tag, fx, l, first + j)
        checked += 1
    return checked


def _check_full_tensors(eng, orc, L, tag):
    """Every materialised map of the images of BLOCKS, EVERY element, against the oracle's maps of the ten source images (the
    oracle is pinned bit for bit to the compiled reference: tests/test_oracle_vs_reference.py) — max-norm and l2, 1e-4."""
    checked = 0
    for l in range(L + 1):
        want = orc.fm(l)                                  # [10, ...]: image i of the batch is source image i % 10
        for first, cnt in BLOCKS:
            try:
                fm = eng.layer_output_range(l, first, cnt)
            except pkg("engine").QcnnError:
                checked -= 1
                break
            for j in range(cnt):
                ref = want[(first + j) % 10]
                d = np.abs(fm[j].astype(np.float64) - ref)
                assert d.max() <= TOL * max(np.abs(ref).max(), 1e-30), "%s: fm[%d] image %d, full tensor max-norm" % (tag, l, first + j)
                assert np.sqrt((d * d).sum()) <= TOL * max(np.sqrt((ref.astype(np.float64) ** 2).sum()), 1e-30), \
                    "%s: fm[%d] image %d, full tensor l2" % (tag, l, first + j)
        checked += 1
    return checked


def _check_outputs(prob, top5, z, fx, tag):
    want = z["prob%d" % fx]
    for i in range(N):                                  # ALL rows of the batch against the ten reference rows (image i = source image i % 10)
        ref = want[i % 10]
        err = np.abs(prob[i].astype(np.float64) - ref).max() / np.abs(ref).max()
        assert err <= TOL, "%s: fixture %d soft-max of image %d: %g" % (tag, fx, i, err)
        t = z["top5_%d" % fx][i % 10]
        if not np.array_equal(top5[i], t):               # a swap is legitimate only between classes the reference itself
            for a, b in zip(top5[i], t):                 # separates by less than the tolerance
                assert a == b or abs(ref[a] - ref[b]) <= TOL * np.abs(ref).max(), "%s: top-5 of image %d" % (tag, i)
    assert np.array_equal(prob[:10], prob[990:1000]) and np.array_equal(top5[:10], top5[990:1000])   # same image, any panel


CONFIGS = {
    # library defaults, every map materialised
    "defaults_layer_for_layer": dict(keep_all=1),
    # what bench.py times: fast path (fused ReLU, fused LRN + pool), one stream, one launch per layer
    "headline_fast_path": dict(keep_all=0, streams=1, host_chunk=0),
    # every eligible layer through the (16-wave) symmetric / sliding kernels whatever the planner would pick
    "forced_sym_slide": dict(keep_all=1, sym=2, slide=2),
    # every eligible layer (conv2 - conv5) through the eight-wave symmetric kernel
    "forced_sym8": dict(keep_all=1, sym8=2),
    # ... and through its sliding form (segments of column strips)
    "forced_sym8_slide": dict(keep_all=0, streams=1, host_chunk=0, sym8=3),
    # every eligible layer through half-panel eight-wave workgroups (qcnn_half8.hip): tile form, and sliding form where built
    "forced_half8": dict(keep_all=1, half8=2),
    "forced_half8_slide": dict(keep_all=0, streams=1, host_chunk=0, half8=3),
    # the north star's scheme for all eight conv / FC layers (bench key value_tables_only)
    "tables_only_fast_path": dict(keep_all=0, streams=1, host_chunk=0, decode=0),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_shipped_parameters_headline_kernels_match_reference(golden_alex_real10, name):
    z = golden_alex_real10
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    L = len(layers)
    imgs = real_bmp_images()[np.arange(N) % 10]
    p1 = synth.load_alexnet_shipped(po.REF_DATA, layers, fixture=1)
    p2 = synth.load_alexnet_shipped(po.REF_DATA, layers, fixture=2)
    eng = _engine(p1, **CONFIGS[name])
    prob, top5 = eng.forward_host(imgs)
    n_maps = _check_maps(eng, z, 1, range(0, L + 1), name)
    _check_outputs(prob, top5, z, 1, name)
    orc = po.COracle(in_chw, layers)
    orc.set_params(p1)
    orc.forward(imgs[:10])
    assert _check_full_tensors(eng, orc, L, name) == n_maps
    if name == "headline_fast_path":
        # which kernels ran (qcnn_get_layer_split: -3 decoded, -5 eight-wave symmetric, -2 sliding): the ones the headline is made of
        assert eng.layer_split(0) == (-3, 2) and eng.layer_split(21)[0] == -3        # conv1 decoded, NCHW in place; fc8 decoded
        assert eng.layer_split(4)[0] == -5 and eng.layer_split(10)[0] == -5      # conv2, conv4: eight-wave symmetric workgroups
        assert eng.layer_split(12)[0] == -2                                      # conv5 sliding
        with pytest.raises(pkg("engine").QcnnError):
            eng.layer_output_range(3, 0, 1)                                      # LRN1 fused into the pool behind it
    if name == "forced_half8":
        assert [eng.layer_split(l)[0] for l in (4, 8, 10, 12)] == [-9, -9, -9, -9]
    if name == "forced_half8_slide":
        assert [eng.layer_split(l)[0] for l in (4, 8, 10, 12)] == [-9, -10, -10, -10]
    if name == "forced_sym8_slide":
        assert [eng.layer_split(l)[0] for l in (4, 8, 10, 12)] == [-6, -6, -6, -6]
    if CONFIGS[name]["keep_all"]:
        assert n_maps == L + 1
    else:
        assert n_maps >= 12
    eng.upload({synth.ALEXNET_FC6: p2[synth.ALEXNET_FC6]})                       # fixture 2: the tail is informative
    prob, top5 = eng.forward_host(imgs)
    assert _check_maps(eng, z, 2, range(16, L + 1), name) >= 4
    _check_outputs(prob, top5, z, 2, name)
    assert len({tuple(t) for t in top5[:10]}) >= 7
    eng.close()
