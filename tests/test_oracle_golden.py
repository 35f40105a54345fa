"""The oracle (oracle/qcnn_oracle.c) against golden vectors produced by the COMPILED REFERENCE
(oracle/make_golden.py).  CPU only; runs wherever the repo goes (no /root/reference needed)."""
import os

import numpy as np
import pytest

import pyoracle as po
from conftest import fingerprint, pkg, tiny_params_from_golden

topo = pkg("topology")
synth = pkg("synth")
SAMPLE_STRIDE = 97


def test_tiny_all_layers_bit_exact(golden_tiny):
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    orc = po.COracle(in_chw, layers)
    orc.set_params(tiny_params_from_golden(z, layers))
    orc.forward(z["imgs"])
    for l in range(len(layers) + 1):
        assert np.array_equal(orc.fm(l), z["fm_%02d" % l]), "fm[%d] differs from the reference" % l
    for i in range(z["imgs"].shape[0]):
        assert np.array_equal(orc.top5(orc.fm(len(layers))[i]), z["top5"][i])


def test_tiny_batch_equals_single(golden_tiny):
    """The reference is batch-1; the oracle's batch loop must not couple images."""
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    orc = po.COracle(in_chw, layers)
    orc.set_params(tiny_params_from_golden(z, layers))
    orc.forward(z["imgs"][1:2])
    one = orc.fm(len(layers))
    assert np.array_equal(one[0], z["fm_%02d" % len(layers)][1])


def test_tiny_layer_isolation(golden_tiny):
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    orc = po.COracle(in_chw, layers)
    orc.set_params(tiny_params_from_golden(z, layers))
    first_fc = [i for i, l in enumerate(layers) if l["type"] == topo.FCNT][0]
    B = z["imgs"].shape[0]
    for l in range(len(layers)):
        x = z["fm_%02d" % l]
        if l == first_fc:
            x = np.ascontiguousarray(x.transpose(0, 3, 1, 2))          # NCHW flatten, src/CaffeEva.cc:187-189
        y = orc.run_layer(l, x, B)
        assert np.array_equal(y, z["fm_%02d" % (l + 1)]), "layer %d" % l


def test_alexnet_conv1_real_parameters(golden_alex_real):
    """conv1 with the SHIPPED codebook/assignments on a real image (stored in full in the fixture)."""
    z = golden_alex_real
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    orc = po.COracle(in_chw, layers)
    orc.set_params({0: dict(bias=z["conv1_bias"], ctrd=z["conv1_ctrd"], asmt=z["conv1_asmt"])})
    y = orc.run_layer(0, z["conv1_in"][None], 1)
    assert np.array_equal(y[0], z["conv1_out"])
    # SURVEY.md §8c fingerprint of fm[1]
    fp = fingerprint(y)
    assert abs(fp[0] - 8.476510e+05) < 1.0 and abs(fp[3] + 2214.58594) < 1e-3 and abs(fp[4] - 2282.12915) < 1e-3


def test_alexnet_synthetic_fingerprints(golden_alex_syn):
    z = golden_alex_syn
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(2, in_chw, seed=8)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    for l in range(len(layers) + 1):
        fm = orc.fm(l)
        smp = fm.reshape(2, -1)[:, ::SAMPLE_STRIDE]
        assert np.array_equal(smp, z["smp_%02d" % l]), "fm[%d] samples" % l
        for i in range(2):
            assert np.allclose(fingerprint(fm[i]), z["fp_%02d" % l][i], rtol=1e-12, atol=0), "fm[%d] fingerprint" % l
    for i in range(2):
        assert np.array_equal(orc.top5(orc.fm(len(layers))[i]), z["top5"][i])


@pytest.mark.skipif(not os.path.isdir(po.REF_DATA), reason="shipped parameters not staged (oracle/_ref/data)")
def test_alexnet_real_fingerprints(golden_alex_real):
    z = golden_alex_real
    in_chw, layers, sub, pfx = topo.MODELS["AlexNet"]
    params = synth.load_param_dir(os.path.join(po.REF_DATA, sub), pfx, layers)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    img = np.ascontiguousarray(z["conv1_in"].transpose(2, 0, 1))[None]     # back to NCHW
    orc.forward(img)
    for l in range(len(layers) + 1):
        fm = orc.fm(l)
        assert np.array_equal(fm.reshape(1, -1)[:, ::SAMPLE_STRIDE], z["smp_%02d" % l]), "fm[%d]" % l
        assert np.allclose(fingerprint(fm), z["fp_%02d" % l][0], rtol=1e-12, atol=0)
    assert np.array_equal(orc.top5(orc.fm(len(layers))[0]), z["top5"][0])
    # SURVEY.md §8c fingerprints
    assert abs(fingerprint(orc.fm(5))[0] + 1.106604e+07) < 10.0
    assert abs(fingerprint(orc.fm(15))[0] - 3.416127e+04) < 0.1
    assert abs(fingerprint(orc.fm(16))[0] - 1.858081e+04) < 0.1


@pytest.mark.skipif(not os.path.isdir(po.REF_DATA), reason="shipped parameters not staged (oracle/_ref/data)")
def test_alexnet_real10_oracle_matches_reference_golden(golden_alex_real10):
    """All ten shipped BMPs, shipped parameters, both fc6 fixtures: the C restatement reproduces bit for bit what the
    COMPILED reference produced (tests/golden/alexnet_real10_ref.npz) — every feature map with fixture 1, the tail
    fm[16..23] with fixture 2 (where fc7 / fc8 / top-5 are not degenerate), soft-max outputs in full, top-5."""
    from conftest import real_bmp_images
    z = golden_alex_real10
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    imgs = real_bmp_images()
    assert np.allclose(np.stack([fingerprint(imgs[i]) for i in range(10)]), z["img_fp"], rtol=1e-12, atol=0)
    L = len(layers)
    for fx in (1, 2):
        orc = po.COracle(in_chw, layers)
        orc.set_params(synth.load_alexnet_shipped(po.REF_DATA, layers, fixture=fx))
        orc.forward(imgs)
        for l in range(16 if fx == 2 else 0, L + 1):
            fm = orc.fm(l).reshape(10, -1)
            assert np.array_equal(fm[:, ::SAMPLE_STRIDE], z["smp%d_%02d" % (fx, l)]), "fixture %d fm[%d]" % (fx, l)
            for i in range(10):
                assert np.allclose(fingerprint(fm[i]), z["fp%d_%02d" % (fx, l)][i], rtol=1e-12, atol=0)
        prob = orc.fm(L).reshape(10, -1)
        assert np.array_equal(prob, z["prob%d" % fx])
        assert np.array_equal(np.stack([orc.top5(prob[i]) for i in range(10)]), z["top5_%d" % fx])
    assert len(set(z["top5_2"][:, 0])) >= 5 and len(set(z["top5_1"][:, 0])) == 1     # fixture 2 is the informative one


def test_precise_path_against_reference_golden():
    """qo_conv_prec / qo_fc_prec (the reference's Init(false) path) against feature maps the COMPILED reference produced for
    the tiny network (tests/golden/tiny_prec_ref.npz, oracle/make_golden.py): bit for bit, strided first layer (the
    im2col quirk at output row / column 0), grouped padded conv and both FC layers included."""
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "tiny_prec_ref.npz"))
    in_chw, layers = topo.tiny_model()
    orc = po.COracle(in_chw, layers)
    orc.set_dense(synth.make_dense_params(in_chw, layers, seed=13))
    orc.forward(z["imgs"])
    for l in range(len(layers) + 1):
        assert np.array_equal(orc.fm(l), z["fm_%02d" % l]), "fm[%d] differs from the reference" % l


def test_perfmodel_sliding_stage_counts():
    """perfmodel.conv_work_slide (what bench.py reports for a layer that ran the sliding kernel) against a brute-force count
    of the (source pixel, sub-space) stages of every (output column, row segment) workgroup; AlexNet conv1 / conv5."""
    perf = pkg("perfmodel")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    sizes = topo.fmap_sizes(in_chw, layers)
    for l, (m, k, cs), segs in ((0, (1, 128, 8), [0, 43, 55]), (12, (24, 128, 8), [0, 13])):
        ly = layers[l]
        wk = perf.conv_work_slide(sizes[l], sizes[l + 1], ly, m, k, cs, segs)
        h, w, _ = sizes[l]
        ho, wo, _ = sizes[l + 1]
        brute = 0
        for x in range(wo):
            cols = [c for c in range(x * ly["stride"] - ly["pad"], x * ly["stride"] - ly["pad"] + ly["knl"]) if 0 <= c < w]
            for a, b in zip(segs[:-1], segs[1:]):
                rows = set()
                for y in range(a, b):
                    rows.update(r for r in range(y * ly["stride"] - ly["pad"], y * ly["stride"] - ly["pad"] + ly["knl"]) if 0 <= r < h)
                brute += len(rows) * len(cols) * m
        assert wk["stages"] == brute * ly["grp"], (l, wk["stages"], brute)
        tile = perf.conv_work(sizes[l], sizes[l + 1], ly, m, k, cs)
        assert wk["stages"] < tile["stages"] and wk["lookups"] == tile["lookups"]


def test_perfmodel_decoded_layer_counts():
    """perfmodel.decoded_report (what bench.py reports for a conv layer that ran through its decoded code words): the matrix
    FLOP issued = outputs x kernel rows x (knl * Cin rounded up to four) x 2, the look-ups it replaces = the reference's
    border-clipped count (35 138 400 for AlexNet conv1)."""
    perf = pkg("perfmodel")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    sizes = topo.fmap_sizes(in_chw, layers)
    r = perf.decoded_report(sizes, layers, 0, 1000.0, 1.8)
    assert r["issued_mfma_flop_per_image"] == 2 * 55 * 55 * 96 * 11 * 36
    assert r["lookups_replaced_per_image"] == 35138400
    assert 0.80 < r["mfma_util"] < 0.82 and r["mfma_util_of_sustained"] > r["mfma_util"]
