"""QCNN_OPT_CHAIN: runs of consecutive conv layers (AlexNet conv3 -> conv4 -> conv5: the reference's plain layer loop,
src/CaffeEva.cc:625-670, with CalcFeatMap_ConvAprx :760-868 per layer) as ONE persistent, dependency-queued launch of the eight-wave
tile kernel (k_conv_chain).  Same body, same tiles, same (kh, kw, m) order per output: BIT-IDENTICAL to one launch per layer —
whatever order the work items of the three layers interleave in."""
import numpy as np
import pytest

import pyoracle as po
from conftest import pkg, rel_err

pytestmark = pytest.mark.gpu

topo = pkg("topology")
synth = pkg("synth")
capi = pkg("capi")
engine = pkg("engine")
TOL = 1e-4


def _engine(in_chw, layers, params, n, chain, streams=1, sym8=2):
    eng = engine.QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)          # the chain is a fast-path feature (ReLU fused into the conv stores)
    eng.set_option(capi.OPT_SPLIT, 0)
    eng.set_option(capi.OPT_STREAMS, streams)
    eng.set_option(capi.OPT_HOST_CHUNK, 0)        # one launch per layer and sub-batch for the whole batch (a host batch would go in two-panel chunks)
    eng.set_option(capi.OPT_HALF8, 0)
    eng.set_option(capi.OPT_SYM8, sym8)           # reference run: the same eight-wave tile kernels, one launch per layer
    eng.set_option(capi.OPT_CHAIN, chain)
    eng.load_model(in_chw, layers, params, n)
    return eng


@pytest.mark.parametrize("n_img,streams", [(300, 1), (1000, 1), (1000, 2)])
def test_conv_chain_alexnet_bit_identical(n_img, streams):
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    params = synth.make_params(in_chw, layers, seed=0)
    imgs = synth.make_images(n_img, in_chw, seed=77)
    base = _engine(in_chw, layers, params, n_img, chain=0, streams=streams)
    p0, t0 = base.forward_host(imgs)
    maps0 = {l: base.layer_output_range(l, n_img - 3, 3) for l in (10, 12, 14)}     # ReLU'd conv3, conv4, conv5 outputs
    base.close()
    eng = _engine(in_chw, layers, params, n_img, chain=1, streams=streams)
    for rep in range(3):                                                           # the interleaving differs from run to run
        p1, t1 = eng.forward_host(imgs)
        assert [eng.layer_split(l) for l in (8, 10, 12)] == [(-11, 1), (-11, 2), (-11, 3)]
        assert eng.layer_split(4)[0] == -5                                          # conv2 (LRN / pool behind it): its own launch
        for l, want in maps0.items():
            assert np.array_equal(eng.layer_output_range(l, n_img - 3, 3), want), "rep %d fm[%d]" % (rep, l)
        assert np.array_equal(p1, p0) and np.array_equal(t1, t0), "rep %d" % rep
    eng.close()


def test_conv_chain_other_geometries_and_the_oracle():
    """A chain of TWO layers behind a pool, with 256 and 384 channels (1x3 and 1x2 tiles), odd maps, a ragged second panel, followed
    by a third conv layer that is NOT eligible (64 channels): chain on = chain off bit for bit, and within 1e-4 of the oracle."""
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.pool(0, 2, 1), topo.conv(1, 3, 256, 1, 1), topo.relu(), topo.conv(1, 3, 384, 2, 1),
              topo.relu(), topo.conv(0, 3, 64, 1, 1), topo.relu(), topo.pool(0, 3, 2), topo.fcnt(40), topo.smax()]
    in_chw = (3, 19, 15)
    params = synth.make_params(in_chw, layers, seed=301)
    imgs = synth.make_images(131, in_chw, seed=302)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[129:])
    outs = []
    for chain in (0, 1):
        eng = _engine(in_chw, layers, params, 131, chain=chain, sym8=1)
        prob, top5 = eng.forward_host(imgs)
        if chain:
            assert [eng.layer_split(l) for l in (3, 5)] == [(-11, 1), (-11, 2)] and eng.layer_split(7)[0] != -11
        outs.append((prob, top5, eng.layer_output(5, 131), eng.layer_output(7, 131)))
        if chain:
            for l in (5, 7, len(layers)):
                e_inf, e_l2 = rel_err(eng.layer_output_range(l, 129, 2), orc.fm(l))
                assert e_inf <= TOL and e_l2 <= TOL, "fm[%d] vs oracle: %g %g" % (l, e_inf, e_l2)
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
