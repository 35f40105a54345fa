"""The launch planner on the CPU (quantized-cnn_amd/csrc/qcnn_planner.{h,hip}: host code, built by g++ into
build/libqcnn_planner_cpu.so — the same functions libqcnn_hip.so plans conv launches with).  Which kernel family runs
GetInPdMat + CalcFeatMap_ConvAprx (src/CaffeEva.cc:1261-1296, :760-868) for a launch geometry is a pure function of plain
numbers; these tests pin the decisions the measured profiles are made of and the properties the decision rules promise."""
import ctypes as C

import pytest

from conftest import pkg

topo = pkg("topology")
synth = pkg("synth")
build = pkg("build")
perf = pkg("perfmodel")

TILE, SLIDE16, SYM16, SYM8, SYM8_SLIDE, HALF8, HALF8_SLIDE = -1, -2, -4, -5, -6, -9, -10
COSTS = ("tile", "slide16", "sym16", "sym8", "sym8_slide", "half8", "half8_slide")


@pytest.fixture(scope="module")
def planner():
    lib = C.CDLL(build.build_planner_cpu())
    lib.qcnn_plan_conv_query.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]

    def query(geom, split=1, slide=1, sym=1, sym8=1, half8=1, lut=1, nchw=0, scratch_mi=64, concurrent=0):
        g = (C.c_int * 14)(*geom)
        o = (C.c_int * 8)(split, slide, sym, sym8, half8, lut, nchw | (concurrent << 1), scratch_mi)
        costs = (C.c_double * 7)()
        ch = (C.c_int * 13)()
        assert lib.qcnn_plan_conv_query(g, o, costs, ch) == 0
        return dict(zip(COSTS, costs)), dict(family=ch[0], split_from=ch[1], Z=ch[2], segs=[ch[4 + i] for i in range(ch[3] + 1)] if ch[3] else [])
    return query


def conv_geoms(model, panels):
    """{layer index: geom[14]} of a model's quantised conv layers (shipped quantisation shapes) for a launch over `panels` panels."""
    in_chw, layers, _, _ = topo.MODELS[model]
    sizes = topo.fmap_sizes(in_chw, layers)
    spec = synth.quant_spec(in_chw, layers)
    out = {}
    for i, l in enumerate(layers):
        if l["type"] == topo.CONV:
            (h, w, c), (ho, wo, ct) = sizes[i], sizes[i + 1]
            out[i] = [h, w, c, ho, wo, ct, l["knl"], l["stride"], l["pad"], l["grp"], spec[i]["M"], spec[i]["Cs"], spec[i]["K"], panels]
    return out


def test_headline_decisions(planner):
    """AlexNet at 1000 images (8 panels), library defaults — the kernels of profiles/r6_*: conv2 and conv4 eight-wave symmetric,
    conv3 half-panel eight-wave, conv5 16-wave sliding; conv1 (one 3-dim sub-space) has no eight-wave form."""
    g = conv_geoms("AlexNet", 8)
    fam = {i: planner(geom)[1]["family"] for i, geom in g.items()}
    assert fam[4] == SYM8 and fam[8] == HALF8 and fam[10] == SYM8 and fam[12] == SLIDE16, fam
    costs0, ch0 = planner(g[0], nchw=1)
    assert ch0["family"] in (TILE, SLIDE16) and costs0["sym8"] == 0 and costs0["half8"] == 0 and costs0["sym16"] == 0


def test_concurrent_sub_batches_prefer_tiles_over_strips(planner):
    """QCNN_OPT_STREAMS = 2 (the library default): the plan covers the panels of both sub-batches, and coarse strips overlap worse than
    tiles — AlexNet conv5 moves from the 16-wave sliding kernel to the eight-wave tile form (measured 9.45 -> 9.36 ms per 1000
    images); VGG-16's layers, where the sliding forms win by 10 - 25 %, keep them."""
    g = conv_geoms("AlexNet", 8)
    assert planner(g[12])[1]["family"] == SLIDE16 and planner(g[12], concurrent=1)[1]["family"] == SYM8
    assert [planner(g[i], concurrent=1)[1]["family"] for i in (4, 8, 10)] == [SYM8, HALF8, SYM8]
    for i, geom in conv_geoms("VGG16", 8).items():
        if geom[5] in (128, 256):
            assert planner(geom, concurrent=1)[1]["family"] == SYM8_SLIDE, i


def test_vgg16_decisions(planner):
    """VGG-16 at 1000 images: the 128- / 256-channel layers keep the eight-wave sliding form, the 512-channel layers run half
    panels (28 x 28 maps: sliding form; 14 x 14 maps: tile or sliding form) — profiles/r6_vgg16."""
    in_chw, layers, _, _ = topo.MODELS["VGG16"]
    g = conv_geoms("VGG16", 8)
    for i, geom in g.items():
        ct = geom[5]
        fam = planner(geom)[1]["family"]
        if ct in (128, 256):
            assert fam == SYM8_SLIDE, (i, fam)
        elif ct == 512:
            assert fam in (HALF8, HALF8_SLIDE), (i, fam)
            if geom[3] == 28:
                assert fam == HALF8_SLIDE, (i, fam)


def test_forced_options_and_modes(planner):
    g = conv_geoms("AlexNet", 8)
    assert [planner(g[i], half8=2)[1]["family"] for i in (4, 8, 10, 12)] == [HALF8] * 4
    assert [planner(g[i], half8=3)[1]["family"] for i in (4, 8, 10, 12)] == [HALF8, HALF8_SLIDE, HALF8_SLIDE, HALF8_SLIDE]   # conv2 (5x5) cannot slide
    assert [planner(g[i], sym8=2, half8=0)[1]["family"] for i in (4, 8, 10, 12)] == [SYM8] * 4
    assert [planner(g[i], sym8=3, half8=0)[1]["family"] for i in (4, 8, 10, 12)] == [SYM8_SLIDE] * 4
    assert planner(g[4], sym=2, sym8=0, half8=0)[1]["family"] == SYM16
    for i in (4, 8, 10, 12):
        # everything off: the tile kernel, whole tiles at 8 panels
        c, ch = planner(g[i], split=0, slide=0, sym=0, sym8=0, half8=0)
        assert ch["family"] == TILE and ch["Z"] <= 1 and all(c[k] == 0 for k in COSTS[1:])
        # the exact builder (LUT mode 0) and the fp16 study modes never take the f32 eight-wave families
        assert planner(g[i], lut=0)[1]["family"] in (TILE, SLIDE16)


def test_one_panel_shard_splits_tiles(planner):
    """One GPU's share of a batch sharded over 8 GPUs (one panel): the 13 x 13 layers cannot fill 256 CUs with whole tiles — the
    planner cuts them (QCNN_OPT_SPLIT), and never does with the split switched off."""
    g = conv_geoms("AlexNet", 1)
    for i in (8, 10, 12):
        _, ch = planner(g[i])
        assert (ch["family"] == TILE and ch["Z"] >= 2) or (ch["family"] == SYM8 and ch["Z"] >= 2), (i, ch)
        _, ch0 = planner(g[i], split=0)
        assert ch0["Z"] <= 1, (i, ch0)


def test_costs_scale_with_the_work(planner):
    """More panels never cost less; at many panels the cost is linear in the panel count (no tail effects left)."""
    for model in ("AlexNet", "VGG16"):
        for i, geom in conv_geoms(model, 1).items():
            prev = None
            for panels in (1, 2, 4, 8, 16, 32):
                geom[13] = panels
                c, _ = planner(geom, split=0)
                for k in COSTS:
                    if prev and prev[k] > 0:
                        assert c[k] >= prev[k] * 0.999, (model, i, k, panels)
                prev = c
            c16 = planner(geom[:13] + [16], split=0)[0]
            c32 = planner(geom[:13] + [32], split=0)[0]
            for k in COSTS:
                if c16[k] > 0:
                    assert (1.7 if "slide" in k else 1.85) <= c32[k] / c16[k] <= 2.05, (model, i, k, c32[k] / c16[k])   # (sliding forms re-cut their segments)


def test_half_panel_tiles_build_less(planner):
    """What the half-panel kernel is for: twice the tile, fewer table builds per source pixel — the planner's tile tables and
    perfmodel.py's restatement agree on the stage counts' ratio for AlexNet conv2 - conv5."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    sizes = topo.fmap_sizes(in_chw, layers)
    spec = synth.quant_spec(in_chw, layers)
    for i in (4, 8, 10, 12):
        w8 = perf.conv_work(sizes[i], sizes[i + 1], layers[i], spec[i]["M"], spec[i]["K"], spec[i]["Cs"], 8)
        wh = perf.conv_work(sizes[i], sizes[i + 1], layers[i], spec[i]["M"], spec[i]["K"], spec[i]["Cs"], "h8")
        assert wh["stages"] / 2 < w8["stages"]            # builds per source pixel and image


def test_malformed_geometry_is_refused(planner):
    lib = C.CDLL(build.build_planner_cpu())
    g = (C.c_int * 14)(13, 13, 256, 13, 13, 385, 3, 1, 1, 2, 32, 8, 128, 8)      # 385 channels in 2 groups
    o = (C.c_int * 8)(1, 1, 1, 1, 1, 1, 0, 64)
    assert lib.qcnn_plan_conv_query(g, o, None, None) != 0
