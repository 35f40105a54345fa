"""The oracle against the compiled reference itself (oracle/_ref/libqcnn_ref.so), live.
CPU only; skipped where oracle/_ref was never built (it is built by __graft_entry__.build() wherever
/root/reference exists and then travels with the repo snapshot)."""
import os

import numpy as np
import pytest

import pyoracle as po
from conftest import pkg

topo = pkg("topology")
synth = pkg("synth")
pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref/libqcnn_ref.so not built")


def _first_fc(layers):
    return [i for i, l in enumerate(layers) if l["type"] == topo.FCNT][0]


def test_custom_topology_end_to_end(tmp_path):
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=21)
    synth.write_param_dir(str(tmp_path), "t", params)
    ref = po.RefLib()
    ref.load_custom(str(tmp_path), "t", in_chw, layers)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    imgs = synth.make_images(2, in_chw, seed=22)
    orc.forward(imgs)
    for i in range(2):
        prob = ref.forward(imgs[i:i + 1])
        for l in range(len(layers) + 1):
            assert np.array_equal(ref.fm(l)[0], orc.fm(l)[i]), "image %d fm[%d]" % (i, l)
        assert np.array_equal(prob, orc.fm(len(layers))[i].reshape(-1))
        assert np.array_equal(ref.top5(), orc.top5(prob))


def test_lut_matches_reference(tmp_path):
    """GetInPdMat output itself (src/CaffeEva.cc:1261-1296), conv (CsEff < Cs) and FC."""
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=23)
    synth.write_param_dir(str(tmp_path), "t", params)
    ref = po.RefLib()
    ref.load_custom(str(tmp_path), "t", in_chw, layers)
    orc = po.COracle(in_chw, layers)
    img = synth.make_images(1, in_chw, seed=24)
    ref.forward(img)
    # conv1: one group -> the table left behind is the whole layer's
    x = ref.fm(0).reshape(-1, in_chw[0])
    m, k, cs = params[0]["ctrd"].shape
    ctrd = np.ascontiguousarray(params[0]["ctrd"].transpose(0, 2, 1))
    lut = np.zeros(x.shape[0] * m * k, np.float32)
    orc.lib.qo_lut_build(np.ascontiguousarray(x), x.shape[0], x.shape[1], ctrd, m, cs, k, lut)
    assert np.array_equal(lut, ref.lut(0))
    fc = _first_fc(layers)
    xin = np.ascontiguousarray(ref.fm(fc).transpose(0, 3, 1, 2)).reshape(1, -1)
    m, k, cs = params[fc]["ctrd"].shape
    ctrd = np.ascontiguousarray(params[fc]["ctrd"].transpose(0, 2, 1))
    lut = np.zeros(m * k, np.float32)
    orc.lib.qo_lut_build(xin, 1, xin.shape[1], ctrd, m, cs, k, lut)
    assert np.array_equal(lut, ref.lut(fc))


def test_single_layers_on_random_activations(tmp_path):
    """SURVEY.md §8c trap: test every layer in isolation on non-degenerate inputs."""
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=25)
    synth.write_param_dir(str(tmp_path), "t", params)
    ref = po.RefLib()
    ref.load_custom(str(tmp_path), "t", in_chw, layers)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    rng = np.random.default_rng(26)
    for l in range(len(layers)):
        h, w, c = orc.fm_dims(l)
        x = (rng.standard_normal((1, h, w, c)) * 3.0).astype(np.float32)
        assert np.array_equal(ref.run_layer(l, x), orc.run_layer(l, x, 1)), "layer %d" % l


@pytest.mark.skipif(not os.path.isdir(po.REF_DATA), reason="shipped parameters not staged")
def test_alexnet_real_bmp_all_feature_maps():
    D = po.REF_DATA
    in_chw, layers, sub, pfx = topo.MODELS["AlexNet"]
    ref = po.RefLib()
    ref.load_named("AlexNet", os.path.join(D, sub), pfx)
    img = ref.load_bmp(os.path.join(D, "AlexNet/imagenet_mean.single.bin"),
                       os.path.join(D, "Bmp.Files/ILSVRC2012_val_00000005.BMP"))
    ref.forward(img)
    orc = po.COracle(in_chw, layers)
    orc.set_params(synth.load_param_dir(os.path.join(D, sub), pfx, layers))
    orc.forward(img)
    for l in range(len(layers) + 1):
        assert np.array_equal(ref.fm(l), orc.fm(l)), "fm[%d]" % l
    assert np.array_equal(ref.top5(), orc.top5(orc.fm(len(layers))[0]))


def test_precise_path_matches_reference(tmp_path):
    """The reference's exact baseline (Init(false): im2col + sgemm, src/CaffeEva.cc:681-758, 932-966, native sgemm
    src/BlasWrapper.cc:55-97): qo_conv_prec / qo_fc_prec against the compiled reference, every feature map of the tiny
    network (grouped conv, padding, stride) bit for bit, and single layers on random activations."""
    in_chw, layers = topo.tiny_model()
    dense = synth.make_dense_params(in_chw, layers, seed=31)
    synth.write_dense_param_dir(str(tmp_path), "t", dense)
    ref = po.RefLib()
    ref.load_custom(str(tmp_path), "t", in_chw, layers, prec=True)
    orc = po.COracle(in_chw, layers)
    orc.set_dense(dense)
    imgs = synth.make_images(2, in_chw, seed=32)
    orc.forward(imgs)
    for i in range(2):
        prob = ref.forward(imgs[i:i + 1])
        for l in range(len(layers) + 1):
            assert np.array_equal(ref.fm(l)[0], orc.fm(l)[i]), "image %d fm[%d]" % (i, l)
        assert np.array_equal(prob, orc.fm(len(layers))[i].reshape(-1))
    rng = np.random.default_rng(33)
    for l, ly in enumerate(layers):
        if ly["type"] in (topo.CONV, topo.FCNT):
            h, w, c = orc.fm_dims(l)
            x = (rng.standard_normal((1, h, w, c)) * 3.0).astype(np.float32)
            assert np.array_equal(ref.run_layer(l, x), orc.run_layer(l, x, 1)), "layer %d" % l
