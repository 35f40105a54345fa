"""GPU parity tests: the HIP path, called through the C-ABI (include/qcnn_hip.h), against the oracle and
the reference-generated golden vectors.

Bars (north_star): conv / FC / ReLU / pool / dropout are integer-indexed fp32 adds in a fixed order —
with the "exact" LUT builder they must be BIT-IDENTICAL to the reference.  LRN and softmax call
expf/logf (device libm vs glibc): <= 1e-6 relative.  MFMA LUT builder (fused multiply-add chain) and
whole-network runs: <= 1e-4 relative, max-norm and l2 (TOL), per feature map.
"""
import os

import numpy as np
import pytest

import pyoracle as po
from conftest import fingerprint, pkg, rel_err, tiny_params_from_golden

pytestmark = pytest.mark.gpu

topo = pkg("topology")
synth = pkg("synth")
fileio = pkg("fileio")
capi = pkg("capi")
TOL = 1e-4          # north_star: "within 1e-4 relative"
TOL_LIBM = 1e-6     # expf/logf differences only
SAMPLE_STRIDE = 97
BITWISE_TYPES = (topo.CONV, topo.FCNT, topo.RELU, topo.POOL, topo.DRPT)


def make_engine(in_chw, layers, params, max_batch, lut=capi.LUT_EXACT, keep_all=1, split=0, decode=0, sym8=0, half8=0):
    """split = 0: one workgroup per tile whatever the batch size (QCNN_OPT_SPLIT off) — the setting under which an image's
    bits do not depend on its batch, which many tests below rely on; the split itself has its own tests.  decode = 0: the
    table kernels for the first layer too (QCNN_OPT_DECODE off) — what these tests are about; the decoded first layer
    has its own tests below and is the default everywhere else (host pipeline, group, bench)."""
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_LUT_MODE, lut)
    eng.set_option(capi.OPT_KEEP_ALL, keep_all)
    eng.set_option(capi.OPT_SPLIT, split)
    eng.set_option(capi.OPT_DECODE, decode)
    eng.set_option(capi.OPT_SYM8, sym8)     # eight-wave symmetric workgroups: their own tests below (default on everywhere else)
    eng.set_option(capi.OPT_HALF8, half8)   # half-panel eight-wave workgroups: tests/test_gpu_half8.py (default: planner)
    eng.load_model(in_chw, layers, params, max_batch)
    return eng


def first_fc(layers):
    return [i for i, l in enumerate(layers) if l["type"] == topo.FCNT][0]


def consumption_order(layers, l, x):
    """fm[l] NHWC -> what layer l consumes (NCHW flatten for the first FC, src/CaffeEva.cc:187-189)."""
    if l == first_fc(layers):
        return np.ascontiguousarray(x.transpose(0, 3, 1, 2))
    return x


# ---------------------------------------------------------------- tiny network, full tensors ----
def test_tiny_layers_in_isolation_exact(golden_tiny):
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    eng = make_engine(in_chw, layers, tiny_params_from_golden(z, layers), 8)
    B = z["imgs"].shape[0]
    for l, ly in enumerate(layers):
        y = eng.run_layer(l, consumption_order(layers, l, z["fm_%02d" % l]), B)
        ref = z["fm_%02d" % (l + 1)]
        if ly["type"] in BITWISE_TYPES:
            assert np.array_equal(y, ref), "layer %d (%s) not bit-identical" % (l, topo.TYPE_NAMES[ly["type"]])
        else:
            e_inf, e_l2 = rel_err(y, ref)
            assert e_inf <= TOL_LIBM and e_l2 <= TOL_LIBM, "layer %d: %g %g" % (l, e_inf, e_l2)


@pytest.mark.parametrize("lut", [capi.LUT_EXACT, capi.LUT_MFMA])
def test_tiny_end_to_end(golden_tiny, lut):
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    eng = make_engine(in_chw, layers, tiny_params_from_golden(z, layers), 8, lut=lut)
    prob, top5 = eng.forward_host(z["imgs"])
    B = z["imgs"].shape[0]
    for l in range(len(layers) + 1):
        e_inf, e_l2 = rel_err(eng.layer_output(l, B), z["fm_%02d" % l])
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d]: %g %g" % (l, e_inf, e_l2)
    assert np.array_equal(top5, z["top5"])
    assert np.allclose(prob, z["fm_%02d" % len(layers)].reshape(B, -1), rtol=TOL, atol=1e-9)
    if lut == capi.LUT_EXACT:      # up to the first libm layer everything is bit-identical
        assert np.array_equal(eng.layer_output(1, B), z["fm_01"])
        assert np.array_equal(eng.layer_output(2, B), z["fm_02"])


def test_tiny_ragged_multi_panel_batch(golden_tiny):
    """130 images = 2 full panels + 2 images: every image must equal its own single-image result."""
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    params = tiny_params_from_golden(z, layers)
    imgs = synth.make_images(130, in_chw, seed=31)
    imgs[:3] = z["imgs"]
    eng = make_engine(in_chw, layers, params, 130)
    eng.set_option(capi.OPT_SMALL_BATCH, 0)      # the batch-size invariance below is a property of the panel kernels
    prob, top5 = eng.forward_host(imgs)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    L = len(layers)
    for l in (1, 4, 5, 7, 8, 11):
        e_inf, e_l2 = rel_err(eng.layer_output(l, 130), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d]: %g %g" % (l, e_inf, e_l2)
    assert np.array_equal(eng.layer_output(1, 130), orc.fm(1))           # conv1, exact builder
    tops = np.stack([orc.top5(orc.fm(L)[i]) for i in range(130)])
    assert np.array_equal(top5, tops)
    assert np.array_equal(top5[:3], z["top5"])
    # batch of 1 and of 64 give the same bits as the batch of 130
    p1, _ = eng.forward_host(imgs[129:130])
    assert np.array_equal(p1[0], prob[129])
    p64, _ = eng.forward_host(imgs[:64])
    assert np.array_equal(p64, prob[:64])


def test_fast_path_equals_layer_for_layer_path(golden_tiny):
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    params = tiny_params_from_golden(z, layers)
    a = make_engine(in_chw, layers, params, 8, lut=capi.LUT_MFMA, keep_all=1)
    b = make_engine(in_chw, layers, params, 8, lut=capi.LUT_MFMA, keep_all=0)
    pa, ta = a.forward_host(z["imgs"])
    pb, tb = b.forward_host(z["imgs"])
    assert np.array_equal(pa, pb) and np.array_equal(ta, tb)
    assert np.array_equal(a.layer_output(2, 3), b.layer_output(2, 3))    # post-ReLU map exists in both
    with pytest.raises(pkg("engine").QcnnError):
        b.layer_output(1, 3)                                             # pre-ReLU map was fused away


def test_error_paths(golden_tiny):
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    eng = make_engine(in_chw, layers, tiny_params_from_golden(z, layers), 4)
    with pytest.raises(pkg("engine").QcnnError):
        eng.forward_host(synth.make_images(5, in_chw))                   # batch > max_batch
    bad = tiny_params_from_golden(z, layers)
    bad[0] = dict(bad[0], asmt=np.full_like(bad[0]["asmt"], 200))        # index >= K
    eng2 = pkg("engine").QcnnEngine(0)
    with pytest.raises(pkg("engine").QcnnError):
        eng2.load_model(in_chw, layers, bad, 4)


# ---------------------------------------------------------------- AlexNet ----
def test_alexnet_conv1_real_parameters_bitwise(golden_alex_real):
    z = golden_alex_real
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=7)
    params[0] = dict(bias=z["conv1_bias"], ctrd=z["conv1_ctrd"], asmt=z["conv1_asmt"])
    eng = make_engine(in_chw, layers, params, 4)
    y = eng.run_layer(0, z["conv1_in"][None], 1)
    assert np.array_equal(y[0], z["conv1_out"])
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA)
    e_inf, e_l2 = rel_err(eng.run_layer(0, z["conv1_in"][None], 1)[0], z["conv1_out"])
    assert e_inf <= TOL and e_l2 <= TOL


@pytest.mark.parametrize("lut", [capi.LUT_EXACT, capi.LUT_MFMA])
def test_alexnet_synthetic_vs_reference_golden(golden_alex_syn, lut):
    z = golden_alex_syn
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(2, in_chw, seed=8)
    eng = make_engine(in_chw, layers, params, 2, lut=lut)
    prob, top5 = eng.forward_host(imgs)
    worst = 0.0
    for l in range(len(layers) + 1):
        fm = eng.layer_output(l, 2)
        smp = fm.reshape(2, -1)[:, ::SAMPLE_STRIDE]
        scale = max(abs(z["fp_%02d" % l][:, 3]).max(), abs(z["fp_%02d" % l][:, 4]).max())
        err = np.abs(smp.astype(np.float64) - z["smp_%02d" % l]).max() / scale
        worst = max(worst, err)
        assert err <= TOL, "fm[%d] samples: %g" % (l, err)
        for i in range(2):
            fp, gp = fingerprint(fm[i]), z["fp_%02d" % l][i]
            assert abs(fp[2] - gp[2]) <= TOL * gp[2], "fm[%d] l2" % l
    assert np.array_equal(top5, z["top5"])
    print("alexnet synthetic lut=%d worst sample error %.3g" % (lut, worst))


def test_alexnet_layers_in_isolation_bitwise():
    """Every conv/FC layer of AlexNet on the oracle's own activations, exact builder: bit-identical."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(3, in_chw, seed=9)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    eng = make_engine(in_chw, layers, params, 3)
    for l, ly in enumerate(layers):
        x = consumption_order(layers, l, orc.fm(l))
        y = eng.run_layer(l, x, 3)
        if ly["type"] in BITWISE_TYPES:
            assert np.array_equal(y, orc.fm(l + 1)), "layer %d (%s)" % (l, topo.TYPE_NAMES[ly["type"]])
        else:
            e_inf, e_l2 = rel_err(y, orc.fm(l + 1))
            assert e_inf <= TOL_LIBM and e_l2 <= TOL_LIBM, "layer %d: %g %g" % (l, e_inf, e_l2)


@pytest.mark.skipif(not os.path.isdir(po.REF_DATA), reason="shipped parameters not staged (oracle/_ref/data)")
def test_alexnet_real_parameters_all_feature_maps(golden_alex_real):
    z = golden_alex_real
    in_chw, layers, sub, pfx = topo.MODELS["AlexNet"]
    params = synth.load_param_dir(os.path.join(po.REF_DATA, sub), pfx, layers)
    img = np.ascontiguousarray(z["conv1_in"].transpose(2, 0, 1))[None]
    for lut in (capi.LUT_EXACT, capi.LUT_MFMA):
        eng = make_engine(in_chw, layers, params, 1, lut=lut)
        prob, top5 = eng.forward_host(img)
        for l in range(len(layers) + 1):
            fm = eng.layer_output(l, 1)
            scale = max(abs(z["fp_%02d" % l][0, 3]), abs(z["fp_%02d" % l][0, 4]), 1e-30)
            err = np.abs(fm.reshape(1, -1)[:, ::SAMPLE_STRIDE].astype(np.float64) - z["smp_%02d" % l]).max() / scale
            assert err <= TOL, "lut %d fm[%d]: %g" % (lut, l, err)
            assert abs(fingerprint(fm)[2] - z["fp_%02d" % l][0, 2]) <= TOL * max(z["fp_%02d" % l][0, 2], 1e-30)
        assert np.array_equal(top5[0], z["top5"][0])
        if lut == capi.LUT_EXACT:
            assert np.array_equal(eng.layer_output(1, 1)[0], z["conv1_out"])


def test_alexnet_full_batch_properties():
    """BASELINE.json configs[1] size (1000 images): size-independent properties instead of a 90 s oracle run:
    batch invariance (bit-identical to the same image in a 64-image batch), permutation equivariance,
    softmax rows sum to 1, and the oracle on a handful of sampled images."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(1000, in_chw, seed=10)
    eng = make_engine(in_chw, layers, params, 1000, lut=capi.LUT_MFMA, keep_all=0)
    prob, top5 = eng.forward_host(imgs)
    assert prob.shape == (1000, 1000) and np.isfinite(prob).all()
    assert np.abs(prob.sum(axis=1) - 1.0).max() < 1e-4
    p64, t64 = eng.forward_host(imgs[936:1000])
    assert np.array_equal(p64, prob[936:1000]) and np.array_equal(t64, top5[936:1000])
    perm = np.random.default_rng(3).permutation(1000)
    pp, tp = eng.forward_host(imgs[perm])
    assert np.array_equal(pp, prob[perm]) and np.array_equal(tp, top5[perm])
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    pick = [0, 63, 64, 511, 999]
    orc.forward(imgs[pick])
    ref = orc.fm(len(layers)).reshape(len(pick), -1)
    for j, i in enumerate(pick):
        e_inf, e_l2 = rel_err(prob[i], ref[j])
        assert e_inf <= TOL and e_l2 <= TOL, "image %d: %g %g" % (i, e_inf, e_l2)
        assert np.array_equal(top5[i], orc.top5(ref[j]))


def test_alexnet_last_ragged_panel_layer_for_layer():
    """Batch of 1000 = 7 full panels + 104 images: every feature map of an image of the LAST (ragged) panel and of
    one in the middle, layer for layer against the oracle (layer-for-layer mode, so every map exists)."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(1000, in_chw, seed=10)
    eng = make_engine(in_chw, layers, params, 1000, lut=capi.LUT_MFMA, keep_all=1)
    eng.forward_host(imgs, want_prob=False, want_top5=False)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    pick = [999, 897, 500]
    orc.forward(imgs[pick])
    for l in range(len(layers) + 1):
        for j, i in enumerate(pick):
            e_inf, e_l2 = rel_err(eng.layer_output_range(l, i, 1)[0], orc.fm(l)[j])
            assert e_inf <= TOL and e_l2 <= TOL, "image %d fm[%d]: %g %g" % (i, l, e_inf, e_l2)
    eng.close()


def test_lrn_pool_fused_equals_separate_kernels():
    """Fast path: LRN + 3x3/2 max-pool run as one kernel once a sub-batch fills the chip (here 17 panels, one stream;
    23x23 and 11x11 maps give partial pool tiles and clipped ceil-mode windows, the last panel is ragged).  The
    pooled map must equal the layer-for-layer run bit for bit; the normalised map is reported as not materialised."""
    layers = [topo.conv(1, 3, 16, 1, 2), topo.relu(), topo.lorn(5, 0.0001, 0.75, 1.0), topo.pool(0, 3, 2),
              topo.conv(1, 3, 24, 1, 1), topo.relu(), topo.lorn(3, 0.0002, 0.75, 2.0), topo.pool(0, 3, 2),
              topo.fcnt(40), topo.smax()]
    in_chw = (3, 46, 46)
    params = synth.make_params(in_chw, layers, seed=31)
    n = 16 * 128 + 77
    imgs = synth.make_images(n, in_chw, seed=32)
    sep = make_engine(in_chw, layers, params, n, lut=capi.LUT_MFMA, keep_all=1)
    sep.forward_host(imgs, want_prob=False, want_top5=False)
    fus = make_engine(in_chw, layers, params, n, lut=capi.LUT_MFMA, keep_all=0)
    fus.set_option(capi.OPT_STREAMS, 1)
    fus.set_option(capi.OPT_HOST_CHUNK, 0)      # one launch per layer: the 17 panels together fill the chip with the fused kernel
    p_f, t_f = fus.forward_host(imgs)
    for l in (4, 8):
        for first in (0, 500, n - 3):
            assert np.array_equal(fus.layer_output_range(l, first, 3), sep.layer_output_range(l, first, 3)), (l, first)
    with pytest.raises(RuntimeError):
        fus.layer_output_range(3, 0, 1)          # the LRN output of a fused pair does not exist
    p_s, t_s = sep.forward_host(imgs)
    assert np.array_equal(p_f, p_s) and np.array_equal(t_f, t_s)
    sep.close(); fus.close()


def test_fallback_glue_kernels():
    """The general LRN kernel (window sizes other than 3 and 5) and the one-thread-per-image soft-max / top-5 (more
    classes than fit an LDS tile) against the oracle."""
    layers = [topo.conv(1, 3, 32, 1, 1), topo.relu(), topo.lorn(7, 0.0002, 0.75, 1.5), topo.pool(0, 2, 2),
              topo.fcnt(1400), topo.smax()]
    in_chw = (3, 8, 8)
    params = synth.make_params(in_chw, layers, seed=77)
    imgs = synth.make_images(131, in_chw, seed=78)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_EXACT)
    prob, top5 = eng.forward_host(imgs)
    for l in range(len(layers) + 1):
        e_inf, e_l2 = rel_err(eng.layer_output(l, 131), orc.fm(l))
        assert e_inf <= TOL_LIBM and e_l2 <= TOL_LIBM, "fm[%d]: %g %g" % (l, e_inf, e_l2)
    L = len(layers)
    assert np.array_equal(top5, np.stack([orc.top5(orc.fm(L)[i]) for i in range(131)]))
    eng.close()


def test_top5_ties_and_softmax_tile():
    """Soft-max / top-5 through the LDS tile (200 classes over 32 class lanes): classes c and c + 100 share their
    bias and assignment rows, so every probability ties with its twin and the first occurrence must win, exactly as
    the reference's sequential sweeps pick it (src/CaffeEva.cc:1173-1188); the soft-max sum is the sequential one."""
    layers = [topo.conv(1, 3, 32, 1, 1), topo.relu(), topo.pool(0, 2, 2), topo.fcnt(200), topo.smax()]
    in_chw = (3, 8, 8)
    params = synth.make_params(in_chw, layers, seed=81)
    fc = params[3]
    fc["bias"][100:] = fc["bias"][:100]
    fc["asmt"][100:] = fc["asmt"][:100]
    imgs = synth.make_images(70, in_chw, seed=82)                      # three blocks of 32 images, the last ragged
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    L = len(layers)
    want = np.stack([orc.top5(orc.fm(L)[i]) for i in range(70)])
    assert np.array_equal(want[:, 1], want[:, 0] + 100) and np.array_equal(want[:, 3], want[:, 2] + 100)   # first, then twin
    for n in (70, 1):
        eng = make_engine(in_chw, layers, params, n, lut=capi.LUT_EXACT)
        prob, top5 = eng.forward_host(imgs[:n])
        assert np.array_equal(prob[:, :100], prob[:, 100:])
        assert np.array_equal(top5, np.stack([orc.top5(prob[i]) for i in range(n)]))   # the sweeps on the device's own bits
        if n > 2:                                                      # panel kernels: bit-identical with the reference
            assert np.array_equal(top5, want[:n])
        e_inf, e_l2 = rel_err(prob, orc.fm(L)[:n])
        assert e_inf <= TOL_LIBM and e_l2 <= TOL_LIBM
        eng.close()


def test_vgg16_two_panel_batch():
    """BASELINE.json configs[3] beyond one panel: 130 images (one full panel + 2).  The oracle needs ~10 s per VGG-16
    image, so: image 129 of the batch must equal the same image run alone bit for bit (batch invariance), and that
    single-image run is checked against the oracle."""
    in_chw, layers, _, _ = topo.MODELS["VGG16"]
    params = synth.make_params(in_chw, layers, seed=51)
    imgs = synth.make_images(130, in_chw, seed=54)
    eng = make_engine(in_chw, layers, params, 130, lut=capi.LUT_MFMA, keep_all=0)
    eng.set_option(capi.OPT_SMALL_BATCH, 0)      # bit-for-bit batch invariance: panel kernels for every batch size
    prob, top5 = eng.forward_host(imgs)
    assert np.isfinite(prob).all() and np.abs(prob.sum(axis=1) - 1.0).max() < 1e-4
    p1, t1 = eng.forward_host(imgs[129:130])
    assert np.array_equal(p1[0], prob[129]) and np.array_equal(t1[0], top5[129])
    p0, t0 = eng.forward_host(imgs[0:1])
    assert np.array_equal(p0[0], prob[0])
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[129:130])
    ref = orc.fm(len(layers)).reshape(-1)
    e_inf, e_l2 = rel_err(prob[129], ref)
    assert e_inf <= TOL and e_l2 <= TOL, "%g %g" % (e_inf, e_l2)
    assert np.array_equal(top5[129], orc.top5(ref))
    eng.close()


# ---------------------------------------------------------------- dispatch coverage ----
def _run_vs_oracle(in_chw, layers, spec_kw, n_img, seed):
    spec = synth.quant_spec(in_chw, layers, **spec_kw)
    params = synth.make_params(in_chw, layers, seed=seed, spec=spec)
    imgs = synth.make_images(n_img, in_chw, seed=seed + 1)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    L = len(layers)
    for lut in (capi.LUT_EXACT, capi.LUT_MFMA):
        eng = make_engine(in_chw, layers, params, n_img, lut=lut)
        prob, top5 = eng.forward_host(imgs)
        for l, ly in enumerate(layers):
            fm = eng.layer_output(l + 1, n_img)
            if lut == capi.LUT_EXACT and ly["type"] in (topo.CONV, topo.FCNT):
                # layer in isolation on the oracle's own input: bit-identical
                y = eng.run_layer(l, consumption_order(layers, l, orc.fm(l)), n_img)
                assert np.array_equal(y, orc.fm(l + 1)), "layer %d (%s) %r" % (l, topo.TYPE_NAMES[ly["type"]], spec_kw)
            e_inf, e_l2 = rel_err(fm, orc.fm(l + 1))
            assert e_inf <= TOL and e_l2 <= TOL, "lut %d fm[%d] %r: %g %g" % (lut, l + 1, spec_kw, e_inf, e_l2)
        tops = np.stack([orc.top5(orc.fm(L)[i]) for i in range(n_img)])
        assert np.array_equal(top5, tops)
        eng.close()


@pytest.mark.parametrize("spec_kw", [
    dict(conv_k=32, conv_cs=4, fc_k=128, fc_cs=8, last_k=16, last_cs=2),   # conv stages of 4 sub-spaces; FC K=128, 2 k-steps
    dict(conv_k=64, conv_cs=8, fc_k=16, fc_cs=4, last_k=32, last_cs=1),    # conv stages of 2 sub-spaces, 2 k-steps
    dict(conv_k=16, conv_cs=2, fc_k=64, fc_cs=2, last_k=128, last_cs=4),   # 8 sub-spaces per stage, Cs < 4
    dict(conv_k=48, conv_cs=8, fc_k=24, fc_cs=4, last_k=10, last_cs=1),    # K without an MFMA builder -> exact builder
])
def test_tiny_other_quantisation_shapes(spec_kw):
    """Quantisation shapes the shipped models do not use (K < 128 conv, K = 128 FC, Cs in {1, 2, 4, 8},
    K that is not a multiple of 16): every template branch of the two hot kernels against the oracle."""
    in_chw, layers = topo.tiny_model()
    _run_vs_oracle(in_chw, layers, spec_kw, 5, seed=41)


def test_vgg_style_channel_counts():
    """3x3/1/1 convolutions with 64/128/256/512/48 output channels (the VGG-16 layer widths, BASELINE.json
    configs[3], on a small map): every conv tile configuration of qk_conv_aprx against the oracle."""
    layers = [topo.conv(1, 3, 64, 1, 1), topo.relu(), topo.conv(1, 3, 128, 1, 1), topo.relu(), topo.pool(0, 2, 2),
              topo.conv(1, 3, 256, 1, 1), topo.relu(), topo.conv(1, 3, 512, 1, 1), topo.relu(), topo.pool(0, 2, 2),
              topo.conv(1, 3, 48, 1, 1), topo.relu(), topo.conv(0, 1, 384, 2, 1), topo.relu(),
              topo.fcnt(400), topo.relu(), topo.drpt(0.5), topo.fcnt(40), topo.smax()]
    _run_vs_oracle((3, 12, 12), layers, {}, 3, seed=43)


def test_conv_geometries_on_a_rectangular_map():
    """Kernel / stride / pad combinations and a non-square input the shipped models do not have (7x7/3 pad 3,
    5x5/2 pad 0 in two groups, 1x1/2, 3x3/1 pad 2 wider than its map edge): the receptive-field geometry of the
    offset program tables (QkProgram) for every one of them, K = 128, against the oracle."""
    layers = [topo.conv(3, 7, 24, 1, 3), topo.relu(), topo.conv(0, 5, 96, 2, 2), topo.relu(),
              topo.conv(0, 1, 16, 1, 2), topo.relu(), topo.conv(2, 3, 200, 1, 1), topo.relu(),
              topo.fcnt(48), topo.smax()]
    _run_vs_oracle((3, 61, 85), layers, {}, 3, seed=47)


# ---------------------------------------------------------------- the other topologies of src/CaffePara.cc ----
@pytest.mark.parametrize("model,n_img", [("CaffeNet", 2), ("VggCnnS", 2), ("CaffeNetFGD", 2), ("CaffeNetFGB", 2),
                                         ("VGG16", 1)])
def test_other_reference_topologies(model, n_img):
    """Every layer table the reference knows besides AlexNet (src/CaffePara.cc:54-237; BASELINE.json configs[3] =
    VGG-16), full size, synthetic parameters in the shipped quantisation layout: all feature maps against the
    oracle (MFMA builder, <= 1e-4) and every conv/FC layer in isolation with the exact builder (bit-identical).
    CaffeNetFGB's 518-way classifier is not a multiple of the reference's 8-way unroll (src/CaffeEva.cc:1008; the
    reference over-runs its output row there); the plain-loop oracle defines the expected values."""
    in_chw, layers, _, _ = topo.MODELS[model]
    params = synth.make_params(in_chw, layers, seed=51)
    imgs = synth.make_images(n_img, in_chw, seed=52)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    L = len(layers)
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA)
    prob, top5 = eng.forward_host(imgs)
    for l in range(L + 1):
        e_inf, e_l2 = rel_err(eng.layer_output(l, n_img), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "%s fm[%d]: %g %g" % (model, l, e_inf, e_l2)
    assert np.array_equal(top5, np.stack([orc.top5(orc.fm(L)[i]) for i in range(n_img)]))
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)
    for l, ly in enumerate(layers):
        if ly["type"] in (topo.CONV, topo.FCNT):
            y = eng.run_layer(l, consumption_order(layers, l, orc.fm(l)), n_img)
            assert np.array_equal(y, orc.fm(l + 1)), "%s layer %d (%s)" % (model, l, topo.TYPE_NAMES[ly["type"]])
    eng.close()


def test_odd_channel_count_is_rejected():
    """Output channels are handled in pairs (one per wave half): an odd count per group is refused with an error,
    never computed wrongly."""
    in_chw = (3, 8, 8)
    layers = [topo.conv(0, 3, 7, 1, 1), topo.relu(), topo.fcnt(10), topo.smax()]
    eng = pkg("engine").QcnnEngine(0)
    with pytest.raises(pkg("engine").QcnnError):
        eng.configure(in_chw, layers, {0: (1, 16, 3), 2: (63, 16, 4)})


def test_option_values_out_of_range_are_refused():
    """qcnn_set_option never clamps: a value a kernel family does not define (QCNN_OPT_SYM8 = 6 was once silently run as 3) is an
    error with a message, and the option keeps its value.  (Code-word counts: up to 256 per sub-space are accepted — above 128
    through pseudo sub-spaces, test_more_than_128_code_words_per_sub_space; more than 256 cannot be named by a uint8 assignment
    and are refused at qcnn_model_set_layer_shape, as the host mirror does at LoadCaffePara: tests/test_host_mirror.py.)"""
    eng = pkg("engine").QcnnEngine(0)
    for opt, bad in ((capi.OPT_SYM8, 4), (capi.OPT_SYM8, 6), (capi.OPT_SYM8, -1), (capi.OPT_SYM, 3), (capi.OPT_SLIDE, 3),
                     (capi.OPT_LUT_MODE, 4), (capi.OPT_LUT_MODE, -1), (capi.OPT_STREAMS, 0), (capi.OPT_STREAMS, 9)):
        with pytest.raises(pkg("engine").QcnnError):
            eng.set_option(opt, bad)
    for opt, ok in ((capi.OPT_SYM8, 3), (capi.OPT_SYM, 2), (capi.OPT_SLIDE, 2), (capi.OPT_LUT_MODE, 3), (capi.OPT_LUT_MODE, 1)):
        eng.set_option(opt, ok)
    layers = [topo.conv(0, 3, 8, 1, 1), topo.relu(), topo.fcnt(10), topo.smax()]
    with pytest.raises(pkg("engine").QcnnError) as ei:
        eng.configure((3, 8, 8), layers, {0: (1, 257, 3), 2: (72, 16, 4)})      # more than a uint8 assignment can name
    assert "K" in str(ei.value)
    eng.close()


def test_more_than_128_code_words_per_sub_space():
    """The reference's uint8 assignments allow up to 256 code words per sub-space (include/FileIO.h:128-166; GetInPdMat has no
    K limit, src/CaffeEva.cc:1261-1296); a LUT stage here holds 128 rows.  Layers with 128 < K <= 256 are cut into pseudo
    sub-spaces of <= 127 code words + one all-zero row over the same dims (an assignment names its code word in one of them, the
    zero row in the others: x + 0 = x, the same sums in the same order) and run the exact-builder kernels in every LUT mode.
    K = 200 (two pseudo sub-spaces), 256 (three), 130; conv with incomplete last sub-space, FC with 2-dim sub-spaces, a 1-dim
    classifier; 5 and 131 images; device-side .cbn decode of 8-bit streams: every conv / FC layer BIT-IDENTICAL to the oracle in
    isolation, every feature map within 1e-4 (LRN / soft-max: libm)."""
    in_chw, layers = topo.tiny_model()
    for spec_kw in (dict(conv_k=200, conv_cs=8, fc_k=256, fc_cs=4, last_k=130, last_cs=1),
                    dict(conv_k=256, conv_cs=4, fc_k=129, fc_cs=2, last_k=255, last_cs=2)):
        _run_vs_oracle(in_chw, layers, spec_kw, 5, seed=141)
    spec = synth.quant_spec(in_chw, layers, conv_k=200, conv_cs=8, fc_k=256, fc_cs=4, last_k=130, last_cs=1)
    params = synth.make_params(in_chw, layers, seed=143, spec=spec)
    imgs = synth.make_images(131, in_chw, seed=144)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[128:131])
    L = len(layers)
    for cbn in (False, True):
        eng = pkg("engine").QcnnEngine(0)
        eng.set_option(capi.OPT_KEEP_ALL, 0)                                  # library defaults otherwise (f32 MFMA mode, fast path)
        eng.load_model(in_chw, layers, params, 131, upload=not cbn)
        if cbn:
            eng.upload_cbn(params)
        prob, top5 = eng.forward_host(imgs)
        e_inf, e_l2 = rel_err(prob[128:131], orc.fm(L).reshape(3, -1))
        assert e_inf <= TOL_LIBM and e_l2 <= TOL_LIBM, "cbn %r: %g %g" % (cbn, e_inf, e_l2)
        assert np.array_equal(top5[128:131], np.stack([orc.top5(orc.fm(L)[i]) for i in range(3)]))
        eng.close()


# ---------------------------------------------------------------- batches of a few images ----
@pytest.mark.parametrize("model,n_img", [("AlexNet", 1), ("AlexNet", 2), ("CaffeNetFGB", 2), ("VggCnnS", 2), ("VGG16", 1)])
def test_small_batch_kernels_vs_oracle(model, n_img):
    """Batches of one or two images run the conv/FC layers with the few-image kernels (qcnn_small.hip: lanes = output
    channels, QCNN_OPT_SMALL_BATCH = 1, the default).  Every feature map against the oracle, and the same batch through
    the panel kernels: equal to rounding."""
    in_chw, layers, _, _ = topo.MODELS[model]
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(n_img, in_chw, seed=9)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    L = len(layers)
    eng = make_engine(in_chw, layers, params, 64, lut=capi.LUT_MFMA)
    prob, top5 = eng.forward_host(imgs)
    for l in range(L + 1):
        e_inf, e_l2 = rel_err(eng.layer_output(l, n_img), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "%s fm[%d]: %g %g" % (model, l, e_inf, e_l2)
    assert np.array_equal(top5, np.stack([orc.top5(orc.fm(L)[i]) for i in range(n_img)]))
    eng.set_option(capi.OPT_SMALL_BATCH, 0)
    prob_panel, top5_panel = eng.forward_host(imgs)
    assert np.array_equal(top5, top5_panel)
    assert np.abs(prob - prob_panel).max() <= 1e-5 * np.abs(prob_panel).max()
    eng.set_option(capi.OPT_KEEP_ALL, 0)                                  # fast path: ReLU fused, input read in place
    eng.set_option(capi.OPT_SMALL_BATCH, 1)
    prob_fast, top5_fast = eng.forward_host(imgs)
    assert np.array_equal(prob_fast, prob) and np.array_equal(top5_fast, top5)
    eng.close()


def test_small_batch_tiny_model_all_layer_types(golden_tiny):
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    eng = make_engine(in_chw, layers, tiny_params_from_golden(z, layers), 8, lut=capi.LUT_MFMA)
    B = 2
    prob, top5 = eng.forward_host(z["imgs"][:B])
    for l in range(len(layers) + 1):
        e_inf, e_l2 = rel_err(eng.layer_output(l, B), z["fm_%02d" % l][:B])
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d]: %g %g" % (l, e_inf, e_l2)
    assert np.array_equal(top5, z["top5"][:B])


# ---------------------------------------------------------------- packed assignments decoded on the device ----
def test_cbn_payload_decoded_on_the_device(golden_tiny):
    """the bit-packed .cbn payload (7-bit conv, 5-/4-bit FC streams; include/FileIO.h:128-166) crosses
    PCIe as it is and is decoded + permuted into the row-offset table by a kernel.  Same bits out as the host-side
    PrepAsmtBuf path, for the all-layer-types network and for AlexNet's real table sizes (several 4096-byte blocks,
    values ending right at block boundaries)."""
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    params = tiny_params_from_golden(z, layers)
    for p in params.values():
        p["bits"] = fileio.min_bits(np.array([p["ctrd"].shape[1] - 1]))
    a = make_engine(in_chw, layers, params, 8, lut=capi.LUT_EXACT)
    b = pkg("engine").QcnnEngine(0)
    b.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)
    shapes = {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()}
    b.configure(in_chw, layers, shapes)
    b.commit(8)
    b.upload_cbn(params)
    pa, ta = a.forward_host(z["imgs"])
    pb, tb = b.forward_host(z["imgs"])
    assert np.array_equal(pa, pb) and np.array_equal(ta, tb)
    for l in range(len(layers) + 1):
        assert np.array_equal(a.layer_output(l, 3), b.layer_output(l, 3)), l
    a.close(); b.close()
    # AlexNet sizes: conv2 (7 bits, 230 400 values = 49.2 blocks), fc7 (5 bits), fc8 (4 bits) as single layers
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=21)
    shapes = {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()}
    e1 = make_engine(in_chw, layers, params, 2, lut=capi.LUT_MFMA)
    e2 = pkg("engine").QcnnEngine(0)
    e2.set_option(capi.OPT_SPLIT, 0)
    e2.set_option(capi.OPT_DECODE, 0)          # like make_engine: the tables themselves are what is compared
    e2.configure(in_chw, layers, shapes)
    e2.commit(2)
    e2.upload_cbn(params)
    imgs = synth.make_images(2, in_chw, seed=22)
    p1, t1 = e1.forward_host(imgs)
    p2, t2 = e2.forward_host(imgs)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2)
    e1.close(); e2.close()


def test_packed_assignment_stream_read_in_place():
    """SURVEY.md §8f-3, second half: for batches of a few images the FC layers read their assignments from the bit-packed
    stream of the reference's .cbn files (file order [Ct][M], 5 / 4 bits, values across byte but never across 4096-byte block
    boundaries), resident beside the byte table, and unpack them in the kernel (QCNN_OPT_PACKED_FC = 1).  Same
    bits as the byte path (QCNN_OPT_PACKED_FC = 0) for 1, 2 and 3 images — AlexNet sizes: fc6 = 9.4 M values in 1441
    blocks —, whether the parameters came as bytes (packed on the host), as a .cbn payload of the layer's own width (kept as
    it is) or of another width (re-packed), and within 1e-4 of the oracle."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=31)
    imgs = synth.make_images(3, in_chw, seed=32)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    L = len(layers)
    shapes = {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()}

    def engine(upload, packed):
        e = pkg("engine").QcnnEngine(0)
        e.set_option(capi.OPT_PACKED_FC, packed)
        e.configure(in_chw, layers, shapes)
        e.commit(4)
        upload(e)
        return e
    wide = {i: dict(p, bits=p["bits"] + 1) if layers[i]["type"] == topo.FCNT else p for i, p in params.items()}   # 6 / 5 bits
    ref = engine(lambda e: e.upload(params), 0)
    variants = [engine(lambda e: e.upload(params), 1), engine(lambda e: e.upload_cbn(params), 1),
                engine(lambda e: e.upload_cbn(wide), 1)]
    for n in (1, 2, 3):
        p0, t0 = ref.forward_host(imgs[:n])
        for k, eng in enumerate(variants):
            p1, t1 = eng.forward_host(imgs[:n])
            assert np.array_equal(p0, p1) and np.array_equal(t0, t1), (n, k)
            for l in (16, 19, 22):
                assert np.array_equal(eng.layer_output(l, n), ref.layer_output(l, n)), (n, k, l)
        e_inf, e_l2 = rel_err(p0, orc.fm(L).reshape(3, -1)[:n])
        assert e_inf <= TOL and e_l2 <= TOL
    for e in [ref] + variants:
        e.close()


def test_cbn_payload_with_an_index_beyond_k_is_rejected():
    in_chw = (3, 8, 8)
    layers = [topo.conv(0, 3, 8, 1, 1), topo.relu(), topo.fcnt(10), topo.smax()]
    spec = synth.quant_spec(in_chw, layers, conv_k=24, fc_k=24)
    params = synth.make_params(in_chw, layers, seed=5, spec=spec)
    params[0] = dict(params[0], asmt=np.full_like(params[0]["asmt"], 30), bits=5)      # 30 fits 5 bits but K = 24
    params[2]["bits"] = 5
    eng = pkg("engine").QcnnEngine(0)
    eng.configure(in_chw, layers, {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()})
    eng.commit(2)
    with pytest.raises(pkg("engine").QcnnError):
        eng.upload_cbn(params)


# ---------------------------------------------------------------- device-side input pipeline ----
@pytest.mark.parametrize("with_mean", [True, False])
def test_u8_input_pipeline_is_bit_identical_to_host_preprocessing(golden_tiny, with_mean):
    """qcnn_forward_u8 (mean subtraction + centre crop on the device, BmpImgIO::RmMeanImg / CropImg
    src/BmpImgIO.cc:180-224) against the same pre-processing done on the host in fp32."""
    import torch
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    c, h, w = in_chw
    hs, ws, n = h + 9, w + 14, 131                                        # ragged batch, odd crop offsets
    rng = np.random.default_rng(61)
    px = rng.integers(0, 256, size=(n, c, hs, ws), dtype=np.uint8)
    mean = (rng.standard_normal((c, hs, ws)) * 20 + 110).astype(np.float32) if with_mean else None
    oy, ox = (hs - h) // 2, (ws - w) // 2
    full = px.astype(np.float32) - (mean[None] if with_mean else np.float32(0))
    host = np.ascontiguousarray(full[:, :, oy:oy + h, ox:ox + w])
    eng = make_engine(in_chw, layers, tiny_params_from_golden(z, layers), n, lut=capi.LUT_MFMA)
    p_ref, t_ref = eng.forward_host(host)
    fm0_ref = eng.layer_output(0, n)
    dev = torch.device("cuda", 0)
    d_px = torch.from_numpy(px).to(dev)
    d_mean = torch.from_numpy(mean).to(dev) if with_mean else None
    d_prob = torch.empty((n, p_ref.shape[1]), dtype=torch.float32, device=dev)
    d_top5 = torch.empty((n, 5), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    eng.forward_u8_dev(d_px.data_ptr(), hs, ws, d_mean.data_ptr() if with_mean else None, n,
                       d_prob.data_ptr(), d_top5.data_ptr())
    pkg("capi").load().qcnn_sync(eng.h)
    assert np.array_equal(eng.layer_output(0, n), fm0_ref)
    assert np.array_equal(d_prob.cpu().numpy(), p_ref)
    assert np.array_equal(d_top5.cpu().numpy().view(np.uint16), t_ref)
    with pytest.raises(pkg("engine").QcnnError):
        eng.forward_u8_dev(d_px.data_ptr(), h - 1, ws, None, n)            # source smaller than the network input


# ---------------------------------------------------------------- fp16 table entries: tolerance study ----
def test_fp16_lut_tolerance_study(golden_alex_syn):
    """BASELINE.json configs[4]: AlexNet with the look-up-table entries rounded to fp16 (accumulation fp32),
    against the reference's fp32 feature maps.  This is a study, not a parity claim: the bar is the fp16
    rounding itself (2^-11 per entry), the measured per-layer errors are printed (LABBOOK.md §5)."""
    z = golden_alex_syn
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(2, in_chw, seed=8)
    eng = make_engine(in_chw, layers, params, 2, lut=capi.LUT_MFMA_F16)
    prob, top5 = eng.forward_host(imgs)
    rows = []
    for l in range(len(layers) + 1):
        fm = eng.layer_output(l, 2)
        smp = fm.reshape(2, -1)[:, ::SAMPLE_STRIDE]
        scale = max(abs(z["fp_%02d" % l][:, 3]).max(), abs(z["fp_%02d" % l][:, 4]).max())
        err = np.abs(smp.astype(np.float64) - z["smp_%02d" % l]).max() / scale
        rows.append((l, err))
        assert err <= 2e-3, "fm[%d]: %g" % (l, err)
    agree = float(np.mean(top5 == z["top5"]))
    print("fp16-rounded LUT, max-norm relative error per feature map: " +
          " ".join("fm%d=%.1e" % r for r in rows if r[1] > 0) + "; top-5 agreement %.2f" % agree)
    assert max(e for _, e in rows) > 1e-6          # the mode really rounds
    assert agree >= 0.8


def test_fp16_table_storage_kernels():
    """BASELINE.json configs[4] for real: QCNN_OPT_LUT_MODE = 2 keeps the tables of conv2 - conv5 and fc6 / fc7 as fp16 in LDS
    (k_conv_sym8 / k_fc_sym8 in their fp16 form: 256-byte rows, one ds_write_b64 per result tile, ds_read_b64 look-ups,
    v_fma_mix_f32 into fp32 sums).  130 images (a full panel + a ragged one), all feature maps:
      * against the SAME mode through the 16-wave kernels (QCNN_OPT_SYM8 = 0), which round the same entries and keep them in f32
        slots: every conv map bit for bit (same entries, same (kh, kw, m) order), fc6 to 1e-6 (the eight-wave FC kernel cuts the
        sub-space axis differently; in isolation fc7 agrees to 1.5e-7 too: scripts/diag/f16_fc_diag.py), the maps behind it to 1e-4;
      * every such layer in isolation against the oracle's qo_study_mode(1, 0) — entries rounded to fp16 (nearest even) when
        stored, fp32 sums — within 5e-5 (not 1e-6: the oracle builds an entry with separately rounded multiply and add,
        src/CaffeEva.cc:1284-1289, the matrix pipe with a fused chain, so a few entries per thousand land on the other side of an
        fp16 rounding boundary: measured 5e-6 ... 1e-5);
      * against the fp32 oracle: the storage rounding is really there (> 1e-5) and is what LABBOOK.md §5 says it costs (< 2e-3)."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    L = len(layers)
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(130, in_chw, seed=8)
    conv = [i for i, l in enumerate(layers) if l["type"] == topo.CONV]
    fcs = [i for i, l in enumerate(layers) if l["type"] == topo.FCNT]
    real = make_engine(in_chw, layers, params, 130, lut=capi.LUT_MFMA_F16, sym8=1)
    prob, top5 = real.forward_host(imgs)
    assert [real.layer_split(l)[0] for l in conv[1:]] == [-7] * (len(conv) - 1), [real.layer_split(l) for l in conv]
    assert [real.layer_split(l)[0] for l in fcs[:2]] == [-7, -7]
    assert real.layer_split(conv[0])[0] != -7 and real.layer_split(fcs[2])[0] != -7   # one 3-dim sub-space / one-dim sub-spaces: f32 slots
    emu = make_engine(in_chw, layers, params, 130, lut=capi.LUT_MFMA_F16, sym8=0)
    prob_e, top5_e = emu.forward_host(imgs)
    for l in range(1, L + 1):
        a, b = real.layer_output(l, 130), emu.layer_output(l, 130)
        if l <= fcs[0]:
            assert np.array_equal(a, b), "fm[%d]: fp16-storage kernels and rounded-f32 kernels differ in bits" % l
        elif l == fcs[0] + 1:                                            # fc6 on bit-identical inputs: the order of the partial sums only
            assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max(), "fm[%d]" % l
        else:                                                            # behind it a 1e-7 input difference moves table entries across
            assert np.abs(a - b).max() <= 3 * TOL * np.abs(b).max(), "fm[%d]" % l   # fp16 rounding boundaries (measured 3e-5 on fc7, 1e-4 on the soft-max)
    emu.close()
    # against the oracle, layer by layer on the fp32 oracle's own input maps (in a whole network a 1e-7 difference in front of a
    # layer already moves some of ITS table entries across fp16 rounding boundaries: measured 1.3e-4 on conv3's map)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[:3])
    ref = [orc.fm(l).copy() for l in range(L + 1)]
    rows = []
    for l in conv[1:] + fcs[:2]:
        x = np.concatenate([ref[l]] * 44)[:130]
        y = real.run_layer(l, consumption_order(layers, l, x), 130)
        assert real.layer_split(l)[0] == -7
        orc.study_mode(True, False)
        want = orc.run_layer(l, consumption_order(layers, l, ref[l]), 3)
        orc.study_mode(False, False)
        e16, _ = rel_err(y[:3], want)
        e32, _ = rel_err(y[:3], ref[l + 1])
        rows.append((l, e16, e32))
        assert e16 <= 5e-5, "layer %d vs the oracle's fp16-storage study mode: %g" % (l, e16)
        assert 1e-5 < e32 <= 2e-3, "layer %d vs the fp32 oracle: %g" % (l, e32)
    print("fp16 table storage, per layer (vs the oracle's study mode / vs fp32): " + " ".join("L%d=%.1e/%.1e" % r for r in rows))
    orc.forward(imgs[127:130])
    e_inf, _ = rel_err(prob[127:130], orc.fm(L).reshape(3, -1))
    assert e_inf <= 5e-3                                                     # the whole network, soft-max outputs against fp32
    real.close()


def test_fp16_sums_kernels_layer_by_layer():
    """The accumulate half of BASELINE.json configs[4]: QCNN_OPT_LUT_MODE = 3 keeps the running sums of conv2 - conv5 / fc6 / fc7 as
    packed fp16 too (k_conv_sym8 / k_fc_sym8 in their fp16-sum form: v_pk_add_f16 after every look-up, twice the tile per wave).
    Every such layer in isolation, on the fp32 oracle's own input maps of 130 images, against the oracle's qo_study_mode(1, 1)
    (entries AND running sums rounded to fp16 after every addition, the bias start value too, same (kh, kw, m) order): the conv
    layers within 2e-3 of the map's largest value (an entry that rounds the other way moves a sum by an fp16 ulp of its
    magnitude); the FC layers — whose sub-space axis is cut over workgroups, the slices added in fp32: another grouping of the
    fp16 roundings, measured 1.7e-2 — and all of them within what LABBOOK.md §5 says fp16 sums cost against fp32 (5e-2).  The whole network in that mode keeps its top-1."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    L = len(layers)
    params = synth.make_params(in_chw, layers, seed=7)
    imgs = synth.make_images(130, in_chw, seed=8)
    conv = [i for i, l in enumerate(layers) if l["type"] == topo.CONV]
    fcs = [i for i, l in enumerate(layers) if l["type"] == topo.FCNT]
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[:3])
    ref = [orc.fm(l).copy() for l in range(L + 1)]
    eng = make_engine(in_chw, layers, params, 130, lut=capi.LUT_MFMA_F16ACC, sym8=1)
    rows = []
    for l in conv[1:] + fcs[:2]:
        x = np.concatenate([ref[l]] * 44)[:130]                              # two panels of the oracle's three images
        y = eng.run_layer(l, consumption_order(layers, l, x), 130)
        assert eng.layer_split(l)[0] == -8, eng.layer_split(l)
        assert np.array_equal(y[:3], y[126:129]) or l in fcs                # same image, any panel (FC: per-launch split)
        orc.study_mode(True, True)
        want = orc.run_layer(l, consumption_order(layers, l, ref[l]), 3)
        orc.study_mode(False, False)
        e16, _ = rel_err(y[:3], want)
        e32, _ = rel_err(y[:3], ref[l + 1])
        rows.append((l, e16, e32))
        assert e16 <= (5e-2 if l in fcs else 2e-3), "layer %d vs the oracle's fp16-sum study mode: %g" % (l, e16)
        assert 1e-4 < e32 <= 5e-2, "layer %d vs fp32: %g" % (l, e32)
    print("fp16 sums, per layer (vs oracle study mode / vs fp32): " + " ".join("L%d=%.1e/%.1e" % r for r in rows))
    prob, top5 = eng.forward_host(imgs)
    orc.forward(imgs[127:130])
    want = orc.fm(L).reshape(3, -1)
    assert np.array_equal(prob[127:130].argmax(axis=1), want.argmax(axis=1))
    assert np.abs(prob[127:130] - want).max() <= 0.1 * want.max()
    eng.close()


def test_result_does_not_depend_on_the_number_of_streams(golden_tiny):
    """QCNN_OPT_STREAMS cuts a forward into sub-batches of whole panels on concurrent HIP streams: same bits."""
    z = golden_tiny
    in_chw, layers = topo.tiny_model()
    params = tiny_params_from_golden(z, layers)
    imgs = synth.make_images(5 * 128 + 7, in_chw, seed=71)               # 6 panels, the last one ragged
    ref = None
    for ns in (1, 2, 3, 4):
        eng = make_engine(in_chw, layers, params, imgs.shape[0], lut=capi.LUT_MFMA, keep_all=0)
        eng.set_option(capi.OPT_STREAMS, ns)
        for _ in range(2):                                                # back-to-back forwards reuse the buffers
            prob, top5 = eng.forward_host(imgs)
        fm = eng.layer_output(4, imgs.shape[0])
        if ref is None:
            ref = (prob, top5, fm)
        assert np.array_equal(prob, ref[0]) and np.array_equal(top5, ref[1]) and np.array_equal(fm, ref[2]), ns
        eng.close()
    with pytest.raises(pkg("engine").QcnnError):
        eng2 = pkg("engine").QcnnEngine(0)
        eng2.set_option(capi.OPT_STREAMS, 9)


# ---------------------------------------------------------------- launches that do not fill the chip: split tiles ----
@pytest.mark.parametrize("n_img", [125, 250])
def test_split_tiles_of_a_sharded_batch(n_img):
    """QCNN_OPT_SPLIT (default on): one GPU's share of a 1000-image batch at 8 / 4 GPUs.  conv3-5 of AlexNet launch fewer
    workgroups than fit whole rounds of 256 CUs, so the tail of their tiles is cut into slices with a fixed-order
    reduction (k_conv_sum) and the FC layers pick their split for the panel count.  Against the same batch without the
    split: every materialised map within 1e-5 (summation order only), same top-5; against the oracle on sampled images:
    within 1e-4.  The exact builder never splits: bit-identical with the option on."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=77)
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    p0, t0 = base.forward_host(imgs)
    fm0 = {l: base.layer_output(l, n_img) for l in (9, 11, 13, 15, 16, 19, 22)}
    base.close()
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=1)
    eng.set_option(capi.OPT_STREAMS, 1)
    p1, t1 = eng.forward_host(imgs)
    cut = {l: eng.layer_split(l) for l in (0, 4, 8, 10, 12)}
    assert any(z > 1 for _, z in cut.values()), cut                      # something was actually split
    for l, want in fm0.items():
        e_inf, _ = rel_err(eng.layer_output(l, n_img), want)
        assert e_inf <= 1e-5, "fm[%d]: %g (split %r)" % (l, e_inf, cut)
    assert np.array_equal(t0, t1)
    assert np.abs(p1 - p0).max() <= 1e-5 * np.abs(p0).max()
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    for i in (0, n_img - 1):
        orc.forward(imgs[i:i + 1])
        e_inf, e_l2 = rel_err(p1[i], orc.fm(len(layers)).reshape(-1))
        assert e_inf <= TOL and e_l2 <= TOL
        e_inf, e_l2 = rel_err(eng.layer_output_range(13, i, 1), orc.fm(13))
        assert e_inf <= TOL and e_l2 <= TOL
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)
    pe, _ = eng.forward_host(imgs[:3])
    assert all(z == 1 for _, z in (eng.layer_split(l) for l in (0, 4, 8, 10, 12)))
    eng.close()


@pytest.mark.parametrize("n_img", [125, 250])
def test_split_tiles_of_the_eight_wave_kernel(n_img):
    """The same split for k_conv_sym8 (QCNN_OPT_SPLIT with the eight-wave tile form): at one or two panels its 2x3 / 1x2 / 2x2
    tiles are 90 - 250 workgroups on 256 CUs, so every tile is cut into Z slices of its (pixel, sub-space) stage sequence, the
    bias enters slice 0, k_conv_sum adds the slices in order.  Forced eight-wave kernels (QCNN_OPT_SYM8 = 2) with and without
    the split: every conv map within 1e-5 (summation order only), same top-5; the last images against the oracle (1e-4); other
    geometries (padded 3x3, 5x5 / 2 in two groups, 192 / 256 / 384 channels) the same way."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=78)
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0, sym8=2)
    p0, t0 = base.forward_host(imgs)
    assert [base.layer_split(l) for l in (4, 8, 10, 12)] == [(-5, 1)] * 4
    fm0 = {l: base.layer_output(l, n_img) for l in (5, 9, 11, 13)}
    base.close()
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=1, sym8=2)
    eng.set_option(capi.OPT_STREAMS, 1)
    p1, t1 = eng.forward_host(imgs)
    cut = {l: eng.layer_split(l) for l in (4, 8, 10, 12)}
    assert all(c[0] == -5 for c in cut.values()) and any(c[1] > 1 for c in cut.values()), cut
    if n_img <= 128:
        assert all(cut[l][1] > 1 for l in (8, 10, 12)), cut                 # one panel of a 13x13 layer: 91 / 98 tiles
    for l, want in fm0.items():
        e_inf, _ = rel_err(eng.layer_output(l, n_img), want)
        assert e_inf <= 1e-5, "fm[%d]: %g (split %r)" % (l, e_inf, cut)
    assert np.array_equal(t0, t1) and np.abs(p1 - p0).max() <= 1e-5 * np.abs(p0).max()
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    for i in (0, n_img - 1):
        orc.forward(imgs[i:i + 1])
        for l in (5, 9, 11, 13):
            e_inf, e_l2 = rel_err(eng.layer_output_range(l, i, 1), orc.fm(l))
            assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] image %d" % (l, i)
    eng.close()
    if n_img > 128:
        return
    geo = ((3, 21, 17), [topo.conv(1, 3, 128, 1, 1), topo.relu(), topo.conv(2, 5, 256, 2, 2), topo.relu(),
                         topo.conv(1, 3, 384, 1, 1), topo.relu(), topo.conv(1, 3, 384, 2, 1), topo.relu(), topo.conv(1, 3, 256, 1, 1),
                         topo.relu(), topo.fcnt(48), topo.smax()])
    g_chw, g_layers = geo
    g_params = synth.make_params(g_chw, g_layers, seed=93)
    g_imgs = synth.make_images(70, g_chw, seed=94)
    outs = []
    for split in (0, 1):
        e = make_engine(g_chw, g_layers, g_params, 70, lut=capi.LUT_MFMA, keep_all=0, split=split, sym8=2)
        outs.append(e.forward_host(g_imgs) + ([e.layer_split(l) for l, ly in enumerate(g_layers) if ly["type"] == topo.CONV],))
        e.close()
    assert any(z > 1 for f, z in outs[1][2] if f == -5), outs[1][2]
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.abs(outs[0][0] - outs[1][0]).max() <= 1e-5 * np.abs(outs[0][0]).max()


def test_split_tiles_small_geometries():
    """The split on small maps and odd shapes (tiny model, conv geometries with borders, K < 128 program-less kernels):
    every slice boundary falls somewhere inside a tile's (pixel, sub-space) sequence; results within 1e-5 of the unsplit run."""
    cases = [(topo.tiny_model(), {}),
             (((3, 61, 85), [topo.conv(3, 7, 24, 1, 3), topo.relu(), topo.conv(0, 5, 96, 2, 2), topo.relu(),
                             topo.conv(2, 3, 200, 1, 1), topo.relu(), topo.fcnt(48), topo.smax()]), {}),
             (topo.tiny_model(), dict(conv_k=32, conv_cs=4, fc_k=128, fc_cs=8, last_k=16, last_cs=2))]
    for (in_chw, layers), spec_kw in cases:
        spec = synth.quant_spec(in_chw, layers, **spec_kw)
        params = synth.make_params(in_chw, layers, seed=91, spec=spec)
        imgs = synth.make_images(130, in_chw, seed=92)
        outs = []
        for split in (0, 1):
            eng = make_engine(in_chw, layers, params, 130, lut=capi.LUT_MFMA, keep_all=0, split=split)
            outs.append(eng.forward_host(imgs))
            eng.close()
        assert np.array_equal(outs[0][1], outs[1][1])
        assert np.abs(outs[0][0] - outs[1][0]).max() <= 1e-5 * np.abs(outs[0][0]).max()


# ---------------------------------------------------------------- decoded first layer ----
def test_decoded_first_layer_alexnet():
    """QCNN_OPT_DECODE (default on): AlexNet conv1 — one sub-space of 3 dims — runs through the code words its assignments
    name on the matrix pipe instead of through look-up tables.  Same parameters, same function: against the table kernels
    within 2e-6 of the map's largest value (both are fp32 sums of the same products, associated differently), against the
    oracle within 1e-4, top-5 equal; the exact builder and the few-image kernels never take the path."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(131, in_chw, seed=83)                     # a full panel + 3 images (one live image tile)
    base = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1)
    p0, t0 = base.forward_host(imgs)
    assert base.layer_split(0)[0] != -3
    fm0 = {l: base.layer_output(l, 131) for l in (1, 2)}
    base.close()
    for keep in (1, 0):
        eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=keep, decode=1)
        p1, t1 = eng.forward_host(imgs)
        assert eng.layer_split(0) == (-3, 1 if keep else 2)        # fast path: the NCHW batch read in place (k_conv_dec_nchw)
        if keep:
            for l, want in fm0.items():
                got = eng.layer_output(l, 131)
                assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max(), "fm[%d]" % l
        assert np.abs(p1 - p0).max() <= 1e-5 * p0.max() and np.array_equal(t0, t1)
        if keep:
            orc = po.COracle(in_chw, layers)
            orc.set_params(params)
            orc.forward(imgs[129:])
            for l in (1, 2, 5):
                e_inf, e_l2 = rel_err(eng.layer_output_range(l, 129, 2), orc.fm(l))
                assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
            p2, t2 = eng.forward_host(imgs[:2])                        # few-image kernels: tables
            assert eng.layer_split(0)[0] != -3
            for n in (5, 16, 17):                                      # <= 16 images: 16-image work items
                p2, t2 = eng.forward_host(imgs[129 - n + 2:131])
                assert eng.layer_split(0) == (-3, 1)
                m = min(n, 2)
                for l in (1, 2):
                    e_inf, e_l2 = rel_err(eng.layer_output_range(l, n - m, m), orc.fm(l)[2 - m:])
                    assert e_inf <= TOL and e_l2 <= TOL, "n = %d fm[%d] vs oracle: %g %g" % (n, l, e_inf, e_l2)
            eng.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)          # the exact builder: tables, the reference's bits
            eng.forward_host(imgs[:5])
            assert eng.layer_split(0)[0] != -3
        eng.close()


@pytest.mark.parametrize("cin,knl,stride,pad,ct", [(3, 3, 1, 1, 64), (3, 5, 2, 0, 32), (1, 3, 1, 1, 32), (4, 7, 3, 2, 96),
                                                   (2, 4, 2, 1, 64), (3, 1, 1, 0, 96), (3, 3, 1, 1, 128), (3, 3, 2, 0, 160),
                                                   (3, 6, 2, 0, 192), (3, 5, 1, 0, 64)])
def test_decoded_first_layer_geometries(cin, knl, stride, pad, ct):
    """Decoded first layers on shapes AlexNet does not have: VGG-16's padded 3x3 / 1 with 64 channels, 32 and 96 channels,
    1, 2 and 4 input channels (kernel rows of 3 ... 28 products, padded to fours), even kernels, 1x1, strides 1-3, padding
    on every side; 200 images (a ragged second panel), all positions of maps whose size is not a multiple of the
    workgroup's positions.  Against the oracle within 1e-4 and against the table kernels within 2e-6."""
    layers = [topo.conv(pad, knl, ct, 1, stride), topo.relu(), topo.pool(0, 2, 2), topo.fcnt(40), topo.smax()]
    in_chw = (cin, 19, 23)
    params = synth.make_params(in_chw, layers, seed=90 + cin)
    rng = np.random.default_rng(91)
    imgs = (rng.integers(0, 256, size=(200,) + in_chw).astype(np.float32) - 120.0)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[126:131])
    base = make_engine(in_chw, layers, params, 200, lut=capi.LUT_MFMA, keep_all=1)
    base.forward_host(imgs)
    want = base.layer_output(1, 200)
    base.close()
    eng = make_engine(in_chw, layers, params, 200, lut=capi.LUT_MFMA, keep_all=1, decode=1)
    prob, top5 = eng.forward_host(imgs)
    assert eng.layer_split(0) == (-3, 1)
    got = eng.layer_output(1, 200)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    for l in range(1, len(layers) + 1):
        e_inf, e_l2 = rel_err(eng.layer_output_range(l, 126, 5), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
    eng.close()



def test_nchw_in_place_input_at_the_very_end_of_an_allocation():
    """k_conv_dec_nchw reads the caller's device buffer in place.  A batch that ends exactly where its hipMalloc'ed region
    ends — 5 images (one image tile, eleven of its sixteen images past the batch), 130 (a ragged second panel whose eight
    image tiles are all launched) and 16 (a full tile whose over-reading positions have no image behind them) — must be
    classified without touching a byte past the region: every image index and every position of an item that holds the
    last image is clamped inside the kernel (scalar offset and per lane), not left to the buffer's range check.  Results
    against the oracle; a read past the region would be a GPU page fault (or garbage in a live lane)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    layers = [topo.conv(0, 11, 96, 1, 4), topo.relu(), topo.pool(0, 3, 2), topo.fcnt(40), topo.smax()]
    in_chw = (3, 67, 71)
    params = synth.make_params(in_chw, layers, seed=301)
    rng = np.random.default_rng(302)
    imgs = (rng.integers(0, 256, size=(130,) + in_chw).astype(np.float32) - 120.0)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    eng = pkg("engine").QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.load_model(in_chw, layers, params, 130)
    import torch
    for n in (5, 16, 130):
        nbytes = n * imgs[0].nbytes
        region = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)            # whole 2 MB pages: the batch ends where the region ends
        base = C.c_void_p()
        assert hip.hipMalloc(C.byref(base), region) == 0
        dev = base.value + region - nbytes
        x = np.ascontiguousarray(imgs[:n])
        assert hip.hipMemcpy(C.c_void_p(dev), x.ctypes.data_as(C.c_void_p), nbytes, 1) == 0
        prob_d = torch.empty((n, 40), dtype=torch.float32, device="cuda:0")
        top5_d = torch.empty((n, 5), dtype=torch.int16, device="cuda:0")
        eng.forward_dev(dev, n, prob_d.data_ptr(), top5_d.data_ptr())
        eng.sync()
        assert eng.layer_split(0) == (-3, 2)                                  # decoded, NCHW in place
        prob = prob_d.cpu().numpy()
        m = min(n, 3)
        orc.forward(imgs[n - m:n])                                            # the batch's last images
        e_inf, e_l2 = rel_err(prob[n - m:], orc.fm(len(layers)).reshape(m, -1))
        assert e_inf <= TOL and e_l2 <= TOL, "n = %d: %g %g" % (n, e_inf, e_l2)
        e_inf, e_l2 = rel_err(eng.layer_output_range(3, n - m, m), orc.fm(3))
        assert e_inf <= TOL and e_l2 <= TOL, "n = %d pool map: %g %g" % (n, e_inf, e_l2)
        assert hip.hipFree(base) == 0
    eng.close()

@pytest.mark.parametrize("cin,knl,stride,ct", [(3, 11, 4, 96), (3, 7, 2, 96), (1, 3, 1, 96), (4, 4, 2, 192), (2, 8, 3, 96), (2, 12, 5, 96)])
def test_decoded_first_layer_reads_nchw_in_place(cin, knl, stride, ct):
    """QCNN_OPT_DIRECT_DEC (default on): on the fast path an unpadded decoded first layer takes its operands straight
    from the caller's NCHW batch (k over the columns of a kernel row: 1, 2 or 3 steps per row, the last lane group
    overlapping the one before it with zero code words) — no pack pass.  Kernel sizes 3 ... 12, 1 - 4 input channels,
    strides 1 - 5, 5 / 70 / 200 images (one ragged 64-image block; a full panel + a ragged one).  Against the packed
    decoded kernel (QCNN_OPT_DIRECT_DEC = 0) within 2e-6 of the map's largest value — the same products summed in another
    order — and against the oracle within 1e-4; device-resident input (qcnn_forward) as well as the host pipeline."""
    layers = [topo.conv(0, knl, ct, 1, stride), topo.relu(), topo.pool(0, 2, 2), topo.fcnt(40), topo.smax()]
    in_chw = (cin, 29, 31)
    params = synth.make_params(in_chw, layers, seed=190 + cin)
    rng = np.random.default_rng(191)
    imgs = (rng.integers(0, 256, size=(200,) + in_chw).astype(np.float32) - 120.0)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    for n in (200, 70, 5, 3, 1):            # (5: one live image tile; 3, 1: the few-image regime — the other layers run its kernels)
        m = min(n, 3)
        orc.forward(imgs[n - m:n])
        outs = {}
        for direct in (0, 1):
            eng = pkg("engine").QcnnEngine(0)
            eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA)
            eng.set_option(capi.OPT_KEEP_ALL, 0)
            eng.set_option(capi.OPT_DIRECT_DEC, direct)
            eng.load_model(in_chw, layers, params, 200)
            prob, top5 = eng.forward_host(imgs[:n])
            if n > capi.SMALL_BATCH_MAX or direct:
                assert eng.layer_split(0) == (-3, 1 + direct)
            else:
                assert eng.layer_split(0)[0] != -3              # few-image table kernel
            outs[direct] = (eng.layer_output(3, n), prob, top5)
            if direct:
                import torch
                x = torch.from_numpy(imgs[:n]).to("cuda:0")
                prob_d = torch.empty((n, 40), dtype=torch.float32, device="cuda:0")
                eng.forward_dev(x.data_ptr(), n, prob_d.data_ptr())
                eng.sync()
                assert np.array_equal(prob_d.cpu().numpy(), prob)
                for l in (3, 4, 5):
                    e_inf, e_l2 = rel_err(eng.layer_output_range(l, n - m, m), orc.fm(l))
                    assert e_inf <= TOL and e_l2 <= TOL, "n = %d fm[%d] vs oracle: %g %g" % (n, l, e_inf, e_l2)
            eng.close()
        assert np.abs(outs[1][0] - outs[0][0]).max() <= (2e-6 if n > capi.SMALL_BATCH_MAX else 1e-5) * np.abs(outs[0][0]).max()
        assert np.abs(outs[1][1] - outs[0][1]).max() <= 1e-5 * outs[0][1].max()
    # a host batch of three panels goes through in chunks of two (QCNN_OPT_HOST_CHUNK default): the first layer then reads
    # the staged NCHW chunks at their panel offsets — same bits as one launch, on one stream and on two
    big = np.concatenate([imgs, imgs[:110]])
    res = []
    for chunk, streams in ((0, 1), (2, 1), (2, 2), (1, 2)):
        eng = pkg("engine").QcnnEngine(0)
        eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA)
        eng.set_option(capi.OPT_KEEP_ALL, 0)
        eng.set_option(capi.OPT_SPLIT, 0)                       # (the planner's cuts may differ with the panels per launch)
        eng.set_option(capi.OPT_HOST_CHUNK, chunk)
        eng.set_option(capi.OPT_STREAMS, streams)
        eng.load_model(in_chw, layers, params, 310)
        prob, top5 = eng.forward_host(big)
        assert eng.layer_split(0) == (-3, 2)
        res.append((eng.layer_output(3, 310), prob, top5))
        eng.close()
    for r in res[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(res[0], r))
    assert np.array_equal(res[0][1][200:], res[0][1][:110])


def test_decoded_first_layer_vgg16_shape():
    """VGG-16's conv1_1 at full size (224x224x3 -> 64 channels, 3x3 / 1, pad 1): the padded decoded kernel with its border
    predicates on every side of a large map; 5 images (16-image work items) and 130 (64-image items, ragged second panel)
    against the oracle (<= 1e-4) and the table kernels (<= 2e-6)."""
    layers = [topo.conv(1, 3, 64, 1, 1), topo.relu(), topo.pool(0, 8, 8), topo.fcnt(40), topo.smax()]
    in_chw = (3, 224, 224)
    params = synth.make_params(in_chw, layers, seed=97)
    rng = np.random.default_rng(98)
    imgs = (rng.integers(0, 256, size=(130,) + in_chw).astype(np.float32) - 120.0)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[127:130])
    base = make_engine(in_chw, layers, params, 130, lut=capi.LUT_MFMA, keep_all=1)
    base.forward_host(imgs)
    want = base.layer_output_range(1, 127, 3)
    base.close()
    eng = make_engine(in_chw, layers, params, 130, lut=capi.LUT_MFMA, keep_all=1, decode=1)
    eng.forward_host(imgs)
    assert eng.layer_split(0) == (-3, 1)
    got = eng.layer_output_range(1, 127, 3)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    for l in range(1, len(layers) + 1):
        e_inf, e_l2 = rel_err(eng.layer_output_range(l, 127, 3), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
    eng.forward_host(imgs[125:130])
    e_inf, e_l2 = rel_err(eng.layer_output_range(1, 2, 3), orc.fm(1))
    assert e_inf <= TOL and e_l2 <= TOL, "5 images: %g %g" % (e_inf, e_l2)
    eng.close()


def test_decoded_classifier():
    """An FC layer whose sub-spaces have one dim (the 1000-way classifier behind fc7: 16 code words of one float) runs
    through its decoded code words: x @ w with w[k][c] = ctrd[k][asmt[k][c]].  1000 channels = 15 blocks of 64 + 40, 1024
    inputs; 200 images (k slices inside the workgroups only), 3 and 1 images (k slices over workgroups too, partial sums
    added in fixed order).  Against the table kernels within 2e-6, against the oracle within 1e-4."""
    layers = [topo.conv(0, 3, 32, 1, 2), topo.relu(), topo.fcnt(1024), topo.relu(), topo.fcnt(1000), topo.smax()]
    in_chw = (3, 9, 9)
    params = synth.make_params(in_chw, layers, seed=95)
    assert params[4]["ctrd"].shape == (1024, 16, 1)
    imgs = synth.make_images(200, in_chw, seed=96)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[:4])
    base = make_engine(in_chw, layers, params, 200, lut=capi.LUT_MFMA, keep_all=1)
    base.forward_host(imgs)
    want = base.layer_output(5, 200)
    base.close()
    eng = make_engine(in_chw, layers, params, 200, lut=capi.LUT_MFMA, keep_all=1, decode=1, split=1)
    for n in (200, 3, 1):
        prob, top5 = eng.forward_host(imgs[:n])
        got = eng.layer_output(5, n)
        assert np.abs(got - want[:n]).max() <= 2e-6 * np.abs(want).max(), n
        assert eng.layer_split(4)[0] == -3 and (n == 200 or eng.layer_split(4)[1] > 1), n   # few images: k slices over workgroups
        m = min(n, 4)
        for l in (5, 6):
            e_inf, e_l2 = rel_err(eng.layer_output(l, m), orc.fm(l)[:m])
            assert e_inf <= TOL and e_l2 <= TOL, "n = %d fm[%d] vs oracle: %g %g" % (n, l, e_inf, e_l2)
    eng.close()
    # QCNN_OPT_SPLIT = 0 promises batch-size-invariant bits: the decoded classifier then never cuts its k axis over
    # workgroups (the cut depends on the panels of the launch), so an image's outputs do not depend on its batch
    inv = make_engine(in_chw, layers, params, 200, lut=capi.LUT_MFMA, keep_all=1, decode=1, split=0)
    inv.set_option(capi.OPT_SMALL_BATCH, 0)
    p200, _ = inv.forward_host(imgs)
    for n in (130, 70, 3, 1):
        pn, _ = inv.forward_host(imgs[:n])
        assert inv.layer_split(4) == (-3, 1)
        assert np.array_equal(pn, p200[:n]), n
    inv.close()


# ---------------------------------------------------------------- symmetric workgroups (128-channel layers) ----
@pytest.mark.parametrize("n_img", [5, 300])
def test_symmetric_workgroups_alexnet(n_img):
    """QCNN_OPT_SYM = 2 (forced): AlexNet conv2 (2 groups x 128 channels, 5x5, 6 sub-spaces of 8 dims) runs k_conv_sym — all
    16 waves build and gather, 8 channels x a 2x2 tile per wave.  Same table entries in the same (kh, kw, m) order per
    output: BIT-IDENTICAL to the tile kernels; conv5 (also 128 channels per group) is eligible too."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=99)
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    p0, t0 = base.forward_host(imgs)
    fm0 = {l: base.layer_output_range(l, n_img - 2, 2) for l in (5, 13, 15)}
    assert base.layer_split(4)[0] != -4
    base.close()
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    eng.set_option(capi.OPT_SLIDE, 0)
    eng.set_option(capi.OPT_SYM, 2)
    p1, t1 = eng.forward_host(imgs)
    assert eng.layer_split(4) == (-4, 1) and eng.layer_split(12) == (-4, 1)
    assert eng.layer_split(8)[0] != -4 and eng.layer_split(10)[0] != -4      # 384 / 192 channels per group: not eligible
    for l, want in fm0.items():
        assert np.array_equal(eng.layer_output_range(l, n_img - 2, 2), want), "fm[%d]" % l
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1)
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)                  # the exact builder: tile kernels
    eng.forward_host(imgs[:5])
    assert eng.layer_split(4)[0] != -4
    eng.close()


def test_symmetric_workgroups_geometries():
    """k_conv_sym on shapes AlexNet does not have: padded 3x3 / 1 and 5x5 / 2 with 128 channels (one group, 8- and 16-
    channel inputs: one and two sub-spaces of 8 dims), a 4-dim sub-space layer (one k-step), an even kernel, odd maps
    (tiles hanging over the border), in two groups — forced on, against the tile kernels (bit-identical) and the oracle."""
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.conv(1, 3, 128, 1, 1), topo.relu(), topo.conv(2, 5, 256, 2, 2),
              topo.relu(), topo.conv(0, 2, 128, 1, 1), topo.relu(), topo.fcnt(40), topo.smax()]
    in_chw = (3, 21, 17)
    params = synth.make_params(in_chw, layers, seed=101)
    imgs = synth.make_images(131, in_chw, seed=102)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[129:])
    base = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    base.forward_host(imgs)
    want = {l: base.layer_output(l, 131) for l in (3, 5, 7)}
    base.close()
    eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0)
    eng.set_option(capi.OPT_SLIDE, 0)
    eng.set_option(capi.OPT_SYM, 2)
    eng.forward_host(imgs)
    assert [eng.layer_split(l)[0] for l in (2, 4, 6)] == [-4, -4, -4]
    for l, w in want.items():
        assert np.array_equal(eng.layer_output(l, 131), w), "fm[%d]" % l
    for l in (3, 5, 7, len(layers)):
        e_inf, e_l2 = rel_err(eng.layer_output_range(l, 129, 2), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
    eng.close()


# ---------------------------------------------------------------- eight-wave symmetric workgroups (256 registers per wave) ----
@pytest.mark.parametrize("n_img,mode", [(5, 2), (300, 2), (5, 3), (300, 3)])
def test_sym8_workgroups_alexnet(n_img, mode):
    """QCNN_OPT_SYM8 = 3: the SLIDING form of the eight-wave kernel wherever it is built — conv2 (5 slots x 1 column x 16 channels
    per wave), conv3 (3 x 1 x 24 in two channel chunks of 192), conv4 (3 x 1 x 24), conv5 (3 slots x 2 columns x 16) sweep segments
    of output columns.  QCNN_OPT_SYM8 = 2 (forced): AlexNet conv2 (128 channels per group: 16 channels x a 2x3
    tile per wave), conv3 (384: 48 x 1x2), conv4 (192: 24 x 2x2) and conv5 (128: 16 x 2x3) run k_conv_sym8 — eight waves of
    256 registers, all of them building and gathering.  Same table entries in the same (kh, kw, m) order per output:
    BIT-IDENTICAL to the tile kernels, layer for layer; conv1 (one 3-dim sub-space) is not eligible."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=199)
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    p0, t0 = base.forward_host(imgs)
    fm0 = {l: base.layer_output_range(l, n_img - 2, 2) for l in (5, 9, 11, 13, 15)}
    assert all(base.layer_split(l)[0] not in (-2, -4, -5) for l in (4, 8, 10, 12))
    base.close()
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0, sym8=mode)
    p1, t1 = eng.forward_host(imgs)
    assert [eng.layer_split(l)[0] for l in (4, 8, 10, 12)] == ([-5, -5, -5, -5] if mode == 2 else [-6, -6, -6, -6]) and eng.layer_split(0)[0] not in (-5, -6)
    if mode == 3:
        assert all(len(eng.layer_segments(l)) >= 2 for l in (4, 10, 12))
    for l, want in fm0.items():
        assert np.array_equal(eng.layer_output_range(l, n_img - 2, 2), want), "fm[%d]" % l
    # fc6 / fc7 ran the eight-wave FC kernel (768 instead of 384 channels per workgroup: another split of the sub-space axis,
    # i.e. another grouping of the partial sums): equal to rounding
    assert eng.layer_split(15)[0] == -5 and eng.layer_split(18)[0] == -5 and eng.layer_split(21)[0] != -5
    assert np.array_equal(t0, t1) and np.abs(p1 - p0).max() <= 1e-5 * np.abs(p0).max()
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)                  # the exact builder: tile kernels
    eng.forward_host(imgs[:5])
    assert eng.layer_split(8)[0] != -5 and eng.layer_split(15)[0] != -5
    eng.close()


@pytest.mark.parametrize("n_img", [5, 131, 1000])
def test_fc_sym8_kernel(n_img):
    """k_fc_sym8 (FC layers with 32 code words of 4 dims: eight waves of 256 registers, 96 channels per wave, offsets through
    LDS-DMA): a 4096 -> 4096 -> 1000-way tail behind a small conv layer (fc6 / fc7 of AlexNet in shape: 6 channel chunks, the
    last one of 256 channels; sub-space axis split over workgroups for few panels) and a 200-channel layer (one chunk, three
    of eight waves with channels) against the 16-wave FC kernel (<= 1e-5: the partial sums are grouped differently) and the
    oracle (<= 1e-4); ragged last panel."""
    layers = [topo.conv(0, 3, 64, 1, 2), topo.relu(), topo.fcnt(4096), topo.relu(), topo.fcnt(200), topo.relu(),
              topo.fcnt(1000), topo.smax()]
    in_chw = (3, 17, 17)
    params = synth.make_params(in_chw, layers, seed=211)
    assert params[2]["ctrd"].shape[1:] == (32, 4) and params[4]["ctrd"].shape[1:] == (32, 4)
    imgs = synth.make_images(n_img, in_chw, seed=212)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    pick = sorted({0, n_img // 2, n_img - 1})
    orc.forward(imgs[pick])
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SMALL_BATCH, 0)
    p0, t0 = base.forward_host(imgs)
    assert base.layer_split(2)[0] != -5
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0, sym8=1)
    eng.set_option(capi.OPT_SMALL_BATCH, 0)
    p1, t1 = eng.forward_host(imgs)
    assert eng.layer_split(2)[0] == -5 and eng.layer_split(4)[0] == -5
    for l in (3, 5, 7):
        a, b = eng.layer_output_range(l, n_img - 1, 1), base.layer_output_range(l, n_img - 1, 1)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), "fm[%d]" % l
    assert np.abs(p1 - p0).max() <= 1e-5 * np.abs(p0).max()
    for j, i in enumerate(pick):
        for l in (3, 5, len(layers)):
            e_inf, e_l2 = rel_err(eng.layer_output_range(l, i, 1)[0], orc.fm(l)[j])
            assert e_inf <= TOL and e_l2 <= TOL, "image %d fm[%d]: %g %g" % (i, l, e_inf, e_l2)
    # batch-size invariance with QCNN_OPT_SPLIT = 0: the same bits for an image whatever its batch
    if n_img == 131:
        p64, _ = eng.forward_host(imgs[:64])
        assert np.array_equal(p64, p1[:64])
    base.close(); eng.close()


def test_sym8_workgroups_geometries():
    """k_conv_sym8 on shapes AlexNet does not have: 256 channels (32 x 1x3 per wave) and 512 (two channel chunks of 256) behind
    8- and 16-channel inputs, a padded 5x5 / 2 layer in two groups of 96 channels (16 x 2x3, six of the eight waves with
    channels), a 4-dim sub-space layer (one k-step), an even kernel, odd maps (tiles hanging over the border), a ragged
    second panel — forced on, against the tile kernels (bit-identical) and the oracle (<= 1e-4)."""
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.conv(1, 3, 256, 1, 1), topo.relu(), topo.conv(2, 5, 192, 2, 2),
              topo.relu(), topo.conv(0, 2, 512, 1, 1), topo.relu(), topo.pool(0, 3, 2), topo.fcnt(40), topo.smax()]
    in_chw = (3, 21, 17)
    spec = synth.quant_spec(in_chw, layers)
    spec[6] = dict(spec[6], Cs=4, M=spec[6]["D"] // 4)                  # its 192 inputs as 48 sub-spaces of 4 dims (one k-step)
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
    for mode in (2, 3):      # 3: the sliding form where it exists (3x3 / 1 with 256 channels, 5x5 / 2 — three slots — in two groups of 96)
        eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0, sym8=mode)
        eng.forward_host(imgs)
        assert [eng.layer_split(l)[0] for l in (2, 4, 6)] == ([-5, -5, -5] if mode == 2 else [-6, -6, -5]) and eng.layer_split(0)[0] not in (-5, -6)
        for l, w in want.items():
            assert np.array_equal(eng.layer_output(l, 131), w), "mode %d fm[%d]" % (mode, l)
        for l in (3, 5, 7, len(layers)):
            e_inf, e_l2 = rel_err(eng.layer_output_range(l, 129, 2), orc.fm(l))
            assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
        eng.close()


def test_sym8_sliding_form_unpadded_and_short_maps():
    """The sliding form on what the other tests do not have: an UNPADDED 3x3 / 1 layer with 128 channels (two-column strips
    on an odd map: the last strip is one column wide), an unpadded 5x5 / 1 layer with 128 channels (five slots), a map with
    fewer than 2 x slots output rows (the planner must fall back to the tile form) — QCNN_OPT_SYM8 = 3 against the tile
    kernels (bit-identical) and the oracle, 131 images."""
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.conv(0, 3, 128, 1, 1), topo.relu(), topo.conv(0, 5, 128, 1, 1),
              topo.relu(), topo.conv(0, 3, 256, 1, 1), topo.relu(), topo.pool(0, 2, 2), topo.fcnt(40), topo.smax()]
    in_chw = (3, 17, 19)                       # maps: 17x19 -> 15x17 -> 11x13 -> 9x11: every layer has >= 2 x slots output rows
    params = synth.make_params(in_chw, layers, seed=211)
    imgs = synth.make_images(131, in_chw, seed=212)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[129:])
    base = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0)
    base.set_option(capi.OPT_SYM, 0)
    base.set_option(capi.OPT_SLIDE, 0)
    base.forward_host(imgs)
    want = {l: base.layer_output(l, 131) for l in (3, 5, 7)}
    base.close()
    eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0, sym8=3)
    eng.forward_host(imgs)
    assert [eng.layer_split(l)[0] for l in (2, 4, 6)] == [-6, -6, -6]
    for l, w in want.items():
        assert np.array_equal(eng.layer_output(l, 131), w), "fm[%d]" % l
    for l in (3, 5, 7, len(layers)):
        e_inf, e_l2 = rel_err(eng.layer_output_range(l, 129, 2), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
    eng.close()
    # a 3x3 layer on a map of 5 output rows (< 2 x 3 slots): no sliding plan, the tile form runs
    layers2 = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.conv(0, 3, 128, 1, 1), topo.relu(), topo.pool(0, 2, 2), topo.fcnt(40), topo.smax()]
    in2 = (3, 7, 9)
    params2 = synth.make_params(in2, layers2, seed=213)
    imgs2 = synth.make_images(20, in2, seed=214)
    outs = []
    for mode in (0, 3):
        e2 = make_engine(in2, layers2, params2, 20, lut=capi.LUT_MFMA, keep_all=1, split=0, sym8=mode)
        e2.forward_host(imgs2)
        if mode == 3:
            assert e2.layer_split(2)[0] == -5
        outs.append(e2.layer_output(3, 20))
        e2.close()
    assert np.array_equal(outs[0], outs[1])


# ---------------------------------------------------------------- sliding-window conv kernels ----
@pytest.mark.parametrize("n_img", [5, 300])
def test_sliding_kernels_alexnet(n_img):
    """QCNN_OPT_SLIDE = 2 (forced): the conv layers of AlexNet run the sliding kernel — a workgroup sweeps the source rows
    under a segment of one output column and builds every source pixel of the strip once.  Against the tile kernels on
    the same batch: BIT-IDENTICAL (the same table entries summed in the same (kh, kw, m) order); against the oracle: within
    1e-4."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=79)
    base = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    p0, t0 = base.forward_host(imgs)
    fm0 = {l: base.layer_output_range(l, n_img - 2, 2) for l in (1, 13, 15, 22)}
    base.close()
    eng = make_engine(in_chw, layers, params, n_img, lut=capi.LUT_MFMA, keep_all=1, split=0)
    eng.set_option(capi.OPT_SLIDE, 2)
    p1, t1 = eng.forward_host(imgs)
    cut = {l: eng.layer_split(l) for l in (0, 4, 8, 10, 12)}
    # forced: every conv layer slides — conv1 (3 slots x 8 channels per wave), conv2 (5 x 6, two channel chunks), conv3
    # (3 x 12, three chunks), conv4 (3 x 12, two chunks), conv5 (3 x 12); the planner itself takes conv1, conv2?, conv5
    assert all(cut[l][0] == -2 for l in (0, 4, 8, 10, 12)), cut
    for l, want in fm0.items():        # same table entries, same (kh, kw, m) order per output: the same bits
        assert np.array_equal(eng.layer_output_range(l, n_img - 2, 2), want), "fm[%d]" % l
    assert np.array_equal(t0, t1) and np.array_equal(p0, p1)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[n_img - 1:])
    for l in (1, 13):
        e_inf, e_l2 = rel_err(eng.layer_output_range(l, n_img - 1, 1), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_EXACT)                  # the exact builder never slides
    eng.forward_host(imgs[:3])
    assert all(eng.layer_split(l)[0] != -2 for l in (0, 12))
    eng.close()


def test_sliding_kernels_geometries():
    """Sliding on shapes AlexNet does not have: padded 3x3 / 1 with 64 and 128 channels (VGG-16's first blocks: 6 and 12
    channels per wave), 5x5 / 2 with padding, 3x3 / 2, a 2x2 / 1 kernel (two slots), narrow maps (segments shorter than the
    window), in two groups — forced on, against the tile kernels (bit-identical) and the oracle (<= 1e-4)."""
    layers = [topo.conv(1, 3, 64, 1, 1), topo.relu(), topo.conv(1, 3, 128, 1, 1), topo.relu(), topo.conv(2, 5, 96, 1, 2),
              topo.relu(), topo.conv(0, 3, 48, 2, 2), topo.relu(), topo.conv(1, 2, 192, 1, 1), topo.relu(),
              topo.fcnt(40), topo.smax()]
    in_chw = (3, 37, 29)
    params = synth.make_params(in_chw, layers, seed=81)
    imgs = synth.make_images(131, in_chw, seed=82)
    base = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0)
    p0, t0 = base.forward_host(imgs)
    maps0 = [base.layer_output_range(l, 128, 3) for l in range(len(layers) + 1)]
    base.close()
    eng = make_engine(in_chw, layers, params, 131, lut=capi.LUT_MFMA, keep_all=1, split=0)
    eng.set_option(capi.OPT_SLIDE, 2)
    p1, t1 = eng.forward_host(imgs)
    slid = [l for l in (0, 2, 4, 6, 8) if eng.layer_split(l)[0] == -2]
    assert len(slid) >= 4, slid
    for l in range(len(layers) + 1):
        assert np.array_equal(eng.layer_output_range(l, 128, 3), maps0[l]), "fm[%d] (slid %r)" % (l, slid)
    assert np.array_equal(t0, t1) and np.array_equal(p0, p1)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[130:])
    e_inf, e_l2 = rel_err(p1[130], orc.fm(len(layers)).reshape(-1))
    assert e_inf <= TOL and e_l2 <= TOL
    eng.close()
