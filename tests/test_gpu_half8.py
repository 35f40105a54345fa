"""Half-panel eight-wave workgroups (k_conv_half8, quantized-cnn_amd/csrc/qcnn_half8.hip; QCNN_OPT_HALF8): the table kernel of
GetInPdMat (src/CaffeEva.cc:1261-1296) + CalcFeatMap_ConvAprx (:760-868) on workgroups of 64 images.  Same table entries added
in the same (kh, kw, m) order per output as every other f32 table kernel: BIT-IDENTICAL to the 16-wave tile kernels, <= 1e-4
from the oracle."""
import numpy as np
import pytest

import pyoracle as po
from conftest import pkg, rel_err
from test_gpu_parity import TOL, make_engine

pytestmark = pytest.mark.gpu

topo = pkg("topology")
synth = pkg("synth")
capi = pkg("capi")


@pytest.mark.parametrize("n_img", [5, 300])
def test_half8_workgroups_alexnet(n_img):
    """QCNN_OPT_HALF8 = 2 (forced): AlexNet conv2 and conv5 (128 channels per group: two wave sets of 4 waves x 32 channels, a 3x4
    tile), conv3 (384: 48 channels per wave, 2x2), conv4 (192: two sets of 4 x 48, 2x4) — against the tile kernels, layer for
    layer; 5 images = one half panel with images, 300 = two full panels and a ragged third one whose upper half is empty."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=199)
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    p0, t0 = base.forward_host(imgs)
    take = min(n_img, 70)                                  # images of both halves of the last panel with images
    fm0 = {l: base.layer_output_range(l, n_img - take, take) for l in (5, 9, 11, 13)}
    base.close()
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0, half8=2)
    p1, t1 = eng.forward_host(imgs)
    assert [eng.layer_split(l) for l in (4, 8, 10, 12)] == [(-9, 1)] * 4 and eng.layer_split(0)[0] != -9
    for l, want in fm0.items():
        assert np.array_equal(eng.layer_output_range(l, n_img - take, take), want), "fm[%d]" % l
    assert np.array_equal(t0, t1) and np.array_equal(p0, p1)
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)      # the exact builder: tile kernels
    eng.forward_host(imgs[:5])
    assert eng.layer_split(8)[0] != -9
    eng.close()


def test_half8_workgroups_geometries():
    """Shapes AlexNet does not have: 256 channels (32 per wave, 2x3 tile) and 512 (64 per wave, 1x3) behind 8- and 16-channel
    inputs, a padded 5x5 / 2 layer in two groups of 192 channels (two wave sets, 2x4), a 4-dim sub-space layer (one k-step), an even
    kernel, odd maps (tiles hanging over the border), a ragged second panel with three images — forced on, against the tile
    kernels (bit-identical, every element of 131 images) and the oracle (<= 1e-4)."""
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.conv(1, 3, 256, 1, 1), topo.relu(), topo.conv(2, 5, 384, 2, 2),
              topo.relu(), topo.conv(0, 2, 512, 1, 1), topo.relu(), topo.pool(0, 3, 2), topo.fcnt(40), topo.smax()]
    in_chw = (3, 21, 17)
    spec = synth.quant_spec(in_chw, layers)
    spec[6] = dict(spec[6], Cs=4, M=spec[6]["D"] // 4)                  # its 384 inputs as 96 sub-spaces of 4 dims (one k-step)
    params = synth.make_params(in_chw, layers, seed=201, spec=spec)
    imgs = synth.make_images(131, in_chw, seed=202)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[129:])
    base = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    base.forward_host(imgs)
    want = {l: base.layer_output(l, 131) for l in (3, 5, 7)}
    base.close()
    eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0, half8=2)
    eng.forward_host(imgs)
    assert [eng.layer_split(l)[0] for l in (2, 4, 6)] == [-9, -9, -9] and eng.layer_split(0)[0] != -9
    for l, w in want.items():
        assert np.array_equal(eng.layer_output(l, 131), w), "fm[%d]" % l
    for l in (3, 5, 7, len(layers)):
        e_inf, e_l2 = rel_err(eng.layer_output_range(l, 129, 2), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
    eng.close()


@pytest.mark.parametrize("n_img", [5, 300])
def test_half8_sliding_form_alexnet(n_img):
    """QCNN_OPT_HALF8 = 3: the SLIDING form wherever it is built (3x3 / 1 layers: conv3 — 3 slots x 1 column x 48 channels per
    wave —, conv4 — 3 x 2, two wave sets of 48 —, conv5 — 3 x 4, two sets of 32), the tile form elsewhere (conv2: 5x5) — against
    the tile kernels, bit for bit."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=199)
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    p0, t0 = base.forward_host(imgs)
    take = min(n_img, 70)
    fm0 = {l: base.layer_output_range(l, n_img - take, take) for l in (5, 9, 11, 13)}
    base.close()
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0, half8=3)
    p1, t1 = eng.forward_host(imgs)
    assert [eng.layer_split(l)[0] for l in (4, 8, 10, 12)] == [-9, -10, -10, -10]
    assert all(len(eng.layer_segments(l)) >= 2 for l in (8, 10, 12))
    for l, want in fm0.items():
        assert np.array_equal(eng.layer_output_range(l, n_img - take, take), want), "fm[%d]" % l
    assert np.array_equal(t0, t1) and np.array_equal(p0, p1)
    eng.close()


def test_half8_sliding_form_geometries():
    """The sliding form on shapes AlexNet does not have: padded and UNPADDED 3x3 / 1 layers with 128 (strips of four columns on
    odd maps: the last strip is narrower), 256 and 512 channels, a 5x5 / 2 layer (three slots) in two groups of 192, a 4-dim
    sub-space layer, a ragged second panel — QCNN_OPT_HALF8 = 3 against the tile kernels (bit-identical) and the oracle."""
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.conv(0, 3, 128, 1, 1), topo.relu(), topo.conv(1, 3, 256, 1, 1), topo.relu(),
              topo.conv(2, 5, 384, 2, 2), topo.relu(), topo.conv(1, 3, 512, 1, 1), topo.relu(), topo.pool(0, 2, 2), topo.fcnt(40), topo.smax()]
    in_chw = (3, 23, 19)                       # maps: 23x19 -> 21x17 -> 21x17 -> 11x9 -> 11x9
    spec = synth.quant_spec(in_chw, layers)
    spec[8] = dict(spec[8], Cs=4, M=spec[8]["D"] // 4)                  # its 384 inputs as 96 sub-spaces of 4 dims (one k-step)
    params = synth.make_params(in_chw, layers, seed=221, spec=spec)
    imgs = synth.make_images(131, in_chw, seed=222)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[129:])
    base = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    base.forward_host(imgs)
    want = {l: base.layer_output(l, 131) for l in (3, 5, 7, 9)}
    base.close()
    eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0, half8=3)
    eng.forward_host(imgs)
    assert [eng.layer_split(l)[0] for l in (2, 4, 6, 8)] == [-10, -10, -10, -10]
    for l, w in want.items():
        assert np.array_equal(eng.layer_output(l, 131), w), "fm[%d]" % l
    for l in (3, 5, 7, 9, len(layers)):
        e_inf, e_l2 = rel_err(eng.layer_output_range(l, 129, 2), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
    eng.close()
