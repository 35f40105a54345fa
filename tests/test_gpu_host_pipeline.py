"""Host-facing entry points added in round 3 (include/qcnn_hip.h, ABI 3): the pipelined batch loop
(qcnn_forward_host_batches — uploads overlapped with the previous batch's layers, the reference's image loop
src/CaffeEva.cc:168-206), chunked large batches, registered host memory, the duplicate-device test rig of the device
group, and the shapes the few-image kernels hand back to the panel kernels."""
import os

import numpy as np
import pytest

import pyoracle as po
from conftest import pkg, rel_err

pytestmark = pytest.mark.gpu

capi = pkg("capi")
topo = pkg("topology")
synth = pkg("synth")
engine = pkg("engine")
TOL = 1e-4


def _engine(in_chw, layers, params, max_batch, keep_all=0, small=0, lut=capi.LUT_MFMA):
    eng = engine.QcnnEngine(0)
    eng.set_option(capi.OPT_LUT_MODE, lut)
    eng.set_option(capi.OPT_KEEP_ALL, keep_all)
    eng.set_option(capi.OPT_SMALL_BATCH, small)
    eng.set_option(capi.OPT_SPLIT, 0)            # bit-for-bit comparisons across batch sizes: one workgroup per tile
    eng.load_model(in_chw, layers, params, max_batch)
    return eng


@pytest.mark.parametrize("pinned", [False, True])
def test_batches_equal_one_forward_per_batch(pinned):
    """qcnn_forward_host_batches: ragged batch sizes, more batches than input buffers, results bit-identical to one
    qcnn_forward_host per batch; from pageable and from registered host memory."""
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=21)
    sizes = [130, 1, 257, 64, 3, 300]
    allimgs = synth.make_images(sum(sizes), in_chw, seed=22)
    if pinned:
        engine.host_register(allimgs)
    try:
        eng = _engine(in_chw, layers, params, max(sizes))
        offs = np.cumsum([0] + sizes)
        batches = [allimgs[offs[i]:offs[i + 1]] for i in range(len(sizes))]
        want = [eng.forward_host(b) for b in batches]
        for _ in range(2):                                   # the second call reuses buffers and events
            prob, top5 = eng.forward_host_batches(batches)
            for i in range(len(sizes)):
                assert np.array_equal(prob[i], want[i][0]), "batch %d" % i
                assert np.array_equal(top5[i], want[i][1]), "batch %d" % i
        p_only, none = eng.forward_host_batches(batches[:2], want_top5=False)
        assert none is None and np.array_equal(p_only[1], want[1][0])
        eng.close()
    finally:
        if pinned:
            engine.host_unregister(allimgs)


def test_large_host_batch_is_chunked_without_changing_results():
    """qcnn_forward_host on the fast path cuts >= 4 panels into two-panel chunks (upload under compute); layer-for-
    layer mode runs one launch.  Same bits either way (panel kernels: an image does not depend on its batch)."""
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=23)
    imgs = synth.make_images(5 * 128 + 77, in_chw, seed=24)           # chunks 256, 256, 205
    keep = _engine(in_chw, layers, params, imgs.shape[0], keep_all=1)
    want = keep.forward_host(imgs)
    keep.close()
    fast = _engine(in_chw, layers, params, imgs.shape[0], keep_all=0)
    got = fast.forward_host(imgs)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    got = fast.forward_host(imgs[: 4 * 128 + 1])                      # remainder below one panel joins the last chunk
    assert np.array_equal(got[0], want[0][: 4 * 128 + 1])
    fast.close()


def test_host_register_errors_are_reported():
    lib = capi.load()
    assert lib.qcnn_host_register(None, 16) != 0
    assert b"empty" in lib.qcnn_last_error(None)


def test_two_ranks_on_one_device(monkeypatch):
    """QCNN_GROUP_ALLOW_DUP=1: two ranks of a device group on device 0 — no RCCL communicator (RCCL refuses two ranks on
    one device), but the per-rank contexts, the arena hand-over, the host threads and the shard arithmetic are the
    production ones.  Sharded result = single-context result bit for bit, single calls and pipelined batches."""
    monkeypatch.setenv("QCNN_GROUP_ALLOW_DUP", "1")
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=25)
    imgs = synth.make_images(300, in_chw, seed=26)
    single = _engine(in_chw, layers, params, 300)
    want = single.forward_host(imgs)
    single.close()
    grp = engine.QcnnDeviceGroup([0, 0])
    assert grp.size == 2
    grp.set_option(capi.OPT_KEEP_ALL, 0)
    grp.set_option(capi.OPT_SMALL_BATCH, 0)
    grp.set_option(capi.OPT_SPLIT, 0)
    grp.load_model(in_chw, layers, params, 300)
    assert grp.shard_bounds(300, 1) == (150, 300)
    for _ in range(2):
        prob, top5 = grp.forward_host(imgs)
        assert np.array_equal(prob, want[0]) and np.array_equal(top5, want[1])
    pb, tb = grp.forward_host_batches([imgs[:300], imgs[:7], imgs[10:141]])
    assert np.array_equal(pb[0], want[0]) and np.array_equal(pb[1], want[0][:7]) and np.array_equal(tb[2], want[1][10:141])
    one, _ = grp.forward_host(imgs[:1])                                  # fewer images than ranks: rank 1 idles
    assert np.array_equal(one, want[0][:1])
    # device-resident form (qcnn_group_forward): every rank's block already on its device, layers enqueued, one sync
    import torch
    x = torch.from_numpy(imgs).to("cuda:0")
    prob_d = torch.zeros((300, want[0].shape[1]), dtype=torch.float32, device="cuda:0")
    top5_d = torch.zeros((300, 5), dtype=torch.int16, device="cuda:0")
    torch.cuda.synchronize()
    b = [grp.shard_bounds(300, r) for r in range(2)]
    grp.forward_dev([x[lo:hi].data_ptr() for lo, hi in b], 300, [prob_d[lo:hi].data_ptr() for lo, hi in b],
                    [top5_d[lo:hi].data_ptr() for lo, hi in b])
    grp.sync()
    assert np.array_equal(prob_d.cpu().numpy(), want[0]) and np.array_equal(top5_d.cpu().numpy().view(np.uint16), want[1])
    grp.close()


def test_reupload_and_second_broadcast_in_the_fp16_table_mode(monkeypatch):
    """the fp16 program tables (QCNN_OPT_LUT_MODE = 2) are built lazily FROM the arena's assignment bytes; a re-upload
    on rank 0 + a second broadcast refills the other ranks' arenas behind their backs — qcnn_model_mark_loaded must drop their
    tables, or ranks >= 1 keep gathering with the OLD assignments.  Two ranks on one device, 300 images (both ranks have work),
    a 256-channel layer (eligible for the eight-wave fp16 form): parameters A, forward, parameters B, broadcast, forward —
    against a single context that only ever saw B.  The broadcast's own checksum check is exercised on the way."""
    monkeypatch.setenv("QCNN_GROUP_ALLOW_DUP", "1")
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.conv(1, 3, 256, 1, 1), topo.relu(), topo.pool(0, 3, 2), topo.fcnt(40), topo.smax()]
    in_chw = (3, 13, 11)
    pa = synth.make_params(in_chw, layers, seed=41)
    pb = synth.make_params(in_chw, layers, seed=42)
    imgs = synth.make_images(300, in_chw, seed=43)
    single = engine.QcnnEngine(0)
    single.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA_F16)
    single.set_option(capi.OPT_KEEP_ALL, 0)
    single.set_option(capi.OPT_SPLIT, 0)
    single.load_model(in_chw, layers, pb, 300)
    want_b = single.forward_host(imgs)
    assert single.layer_split(2)[0] == -7                                # the eight-wave fp16-table kernel ran
    sum_b = single.arena_checksum()
    single.close()
    grp = engine.QcnnDeviceGroup([0, 0])
    grp.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA_F16)
    grp.set_option(capi.OPT_KEEP_ALL, 0)
    grp.set_option(capi.OPT_SPLIT, 0)
    grp.load_model(in_chw, layers, pa, 300)
    sum_a = grp.arena_checksum()
    prob_a, _ = grp.forward_host(imgs)                                   # builds every rank's fp16 program tables from A
    grp.upload(pb)
    with pytest.raises(engine.QcnnError):
        grp.forward_host(imgs)                                           # a forward between upload and broadcast is refused
    grp.broadcast()
    assert grp.arena_checksum() == sum_b and sum_a != sum_b              # every rank holds B's bytes (a single context's arena of B)
    prob, top5 = grp.forward_host(imgs)
    assert np.array_equal(prob, want_b[0]) and np.array_equal(top5, want_b[1])
    assert not np.array_equal(prob_a, prob)
    grp.close()


def test_group_kernel_family_follows_the_global_batch(monkeypatch):
    """A shard of a few images of a LARGER batch must not take the few-image kernels (bits would then depend on
    the number of GPUs): 4 images (> QCNN_SMALL_BATCH_MAX) over 2 ranks = shards of 2 images, panel kernels on both."""
    monkeypatch.setenv("QCNN_GROUP_ALLOW_DUP", "1")
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=27)
    imgs = synth.make_images(4, in_chw, seed=28)
    single = _engine(in_chw, layers, params, 8, small=0)
    want = single.forward_host(imgs)
    single.close()
    grp = engine.QcnnDeviceGroup([0, 0])
    grp.set_option(capi.OPT_KEEP_ALL, 0)                                 # small-batch option left at its default (on)
    grp.set_option(capi.OPT_SPLIT, 0)
    grp.load_model(in_chw, layers, params, 8)
    prob, top5 = grp.forward_host(imgs)
    assert np.array_equal(prob, want[0]) and np.array_equal(top5, want[1])
    grp.close()


def test_few_image_batches_fall_back_to_the_panel_kernels_where_needed():
    """Shapes the few-image kernels do not cover — an FC code book whose K is not a multiple of 4, a conv window too
    large for their LDS table (15x15 taps, K = 128) — run the panel kernels layer by layer instead of failing."""
    in_chw = (3, 20, 20)
    layers = [topo.conv(0, 15, 16, 1, 1), topo.relu(), topo.fcnt(24), topo.relu(), topo.fcnt(10), topo.smax()]
    spec = synth.quant_spec(in_chw, layers, fc_k=10, fc_cs=4, last_k=16, last_cs=1)
    params = synth.make_params(in_chw, layers, seed=29, spec=spec)
    for n in (1, 2):
        imgs = synth.make_images(n, in_chw, seed=30 + n)
        orc = po.COracle(in_chw, layers)
        orc.set_params(params)
        orc.forward(imgs)
        eng = _engine(in_chw, layers, params, 4, keep_all=1, small=1)
        prob, top5 = eng.forward_host(imgs)
        for l in range(len(layers) + 1):
            e_inf, e_l2 = rel_err(eng.layer_output(l, n), orc.fm(l))
            assert e_inf <= TOL and e_l2 <= TOL, "n=%d fm[%d]: %g %g" % (n, l, e_inf, e_l2)
        eng.close()


def test_first_layer_with_two_subspaces_is_not_read_in_place():
    """A first conv layer whose <= 4 input channels are split into TWO sub-spaces (Cs = 2): the in-place NCHW read is
    built for one sub-space per pixel, so the fast path must pack this input like any other — results against the
    oracle, last image of the batch included (the one whose over-read would leave the caller's buffer)."""
    in_chw = (4, 12, 12)
    layers = [topo.conv(1, 3, 16, 1, 1), topo.relu(), topo.fcnt(10), topo.smax()]
    spec = synth.quant_spec(in_chw, layers, conv_k=128, conv_cs=2)
    assert spec[0]["M"] == 2
    params = synth.make_params(in_chw, layers, seed=33, spec=spec)
    imgs = (np.random.default_rng(34).integers(0, 256, size=(5,) + in_chw).astype(np.float32) - 110.0)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    eng = _engine(in_chw, layers, params, 5, keep_all=0, small=0)
    prob, _ = eng.forward_host(imgs)
    e_inf, e_l2 = rel_err(prob, orc.fm(len(layers)).reshape(5, -1))
    assert e_inf <= TOL and e_l2 <= TOL
    eng.close()


# ---------------------------------------------------------------- the reference's precise path on the device ----
def _dense_vs_oracle(in_chw, layers, n, seed):
    dense = synth.make_dense_params(in_chw, layers, seed=seed)
    imgs = synth.make_images(n, in_chw, seed=seed + 1)
    orc = po.COracle(in_chw, layers)
    orc.set_dense(dense)
    orc.forward(imgs)
    for keep_all in (1, 0):
        eng = engine.QcnnEngine(0)
        eng.set_option(capi.OPT_KEEP_ALL, keep_all)
        eng.load_dense_model(in_chw, layers, dense, n)
        prob, top5 = eng.forward_host(imgs)
        if keep_all:
            for l in range(len(layers) + 1):
                e_inf, e_l2 = rel_err(eng.layer_output(l, n), orc.fm(l))
                assert e_inf <= TOL and e_l2 <= TOL, "fm[%d]: %g %g" % (l, e_inf, e_l2)
        e_inf, e_l2 = rel_err(prob, orc.fm(len(layers)).reshape(n, -1))
        assert e_inf <= TOL and e_l2 <= TOL
        assert np.array_equal(top5, np.stack([orc.top5(orc.fm(len(layers))[i]) for i in range(n)]))
        eng.close()


def test_precise_path_tiny_network():
    """Init(false) of the reference (im2col + sgemm, src/CaffeEva.cc:681-758, 932-966) as dense layers on the device:
    every feature map of the tiny network (strided first layer — the reference's im2col drops some taps at output row /
    column 0 there, reproduced —, grouped padded conv, two FC layers) against the oracle, which is pinned bit for bit
    to the compiled reference (tests/test_oracle_vs_reference.py); 131 images = two panels, the second ragged."""
    in_chw, layers = topo.tiny_model()
    _dense_vs_oracle(in_chw, layers, 131, seed=41)


def test_precise_path_conv_geometries():
    layers = [topo.conv(3, 7, 24, 1, 3), topo.relu(), topo.conv(0, 5, 96, 2, 2), topo.relu(),
              topo.conv(0, 1, 16, 1, 2), topo.relu(), topo.conv(2, 3, 200, 1, 1), topo.relu(),
              topo.fcnt(48), topo.smax()]
    _dense_vs_oracle((3, 61, 85), layers, 3, seed=43)


def test_precise_path_alexnet_first_layers():
    """AlexNet's conv1 (11x11 stride 4: three of every eleven taps miss output row / column 0 in the reference), LRN, pool
    and the grouped conv2 at full size, one image, against the oracle; the approximate and the precise layer types can be
    mixed in one model (conv1 dense, conv2 quantised)."""
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    layers = layers[:6]                                    # conv1 relu lrn pool conv2 relu
    dense = synth.make_dense_params(in_chw, layers, seed=45)
    imgs = synth.make_images(1, in_chw, seed=46)
    orc = po.COracle(in_chw, layers)
    orc.set_dense(dense)
    orc.forward(imgs)
    eng = engine.QcnnEngine(0)
    eng.load_dense_model(in_chw, layers, dense, 1)
    eng.forward_host(imgs, want_top5=False)
    for l in (1, 4, 5):
        e_inf, e_l2 = rel_err(eng.layer_output(l, 1), orc.fm(l))
        assert e_inf <= TOL and e_l2 <= TOL, "fm[%d]: %g %g" % (l, e_inf, e_l2)
    eng.close()
    quant = synth.make_params(in_chw, layers, seed=47)
    mixed = po.COracle(in_chw, layers)
    mixed.set_dense({0: dense[0]})
    mixed.set_params({4: quant[4]})
    mixed.forward(imgs)
    eng = engine.QcnnEngine(0)
    lib = eng.lib
    arr = (capi.QcnnLayerDesc * len(layers))(*[capi.layer_desc(l) for l in layers])
    eng._chk(lib.qcnn_model_begin(eng.h, len(layers), arr, *in_chw))
    eng._chk(lib.qcnn_model_set_layer_dense(eng.h, 0))
    m, k, cs = quant[4]["ctrd"].shape
    eng._chk(lib.qcnn_model_set_layer_shape(eng.h, 4, m, k, cs))
    eng.layers, eng.L, eng.in_chw = layers, len(layers), in_chw
    eng.commit(1)
    w = np.ascontiguousarray(dense[0]["weights"], np.float32)
    b = np.ascontiguousarray(dense[0]["bias"], np.float32)
    eng._chk(lib.qcnn_model_set_layer_weights(eng.h, 0, b.ctypes.data, w.ctypes.data))
    eng.upload({4: quant[4]})
    eng.forward_host(imgs, want_top5=False)
    e_inf, e_l2 = rel_err(eng.layer_output(5, 1), mixed.fm(5))
    assert e_inf <= TOL and e_l2 <= TOL
    eng.close()
