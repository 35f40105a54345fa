"""Device group (qcnn_group_* of include/qcnn_hip.h): one batch sharded over the visible GPUs of this process,
rank 0's parameter arena broadcast with RCCL.  With one visible GPU the group has one rank (the communicator,
the broadcast and the host thread path still run); with two or more the sharded result must equal the
single-context result bit for bit (images are independent)."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu

capi = pkg("capi")
topo = pkg("topology")
synth = pkg("synth")
engine = pkg("engine")


def _single(in_chw, layers, params, imgs):
    eng = engine.QcnnEngine(0)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.set_option(capi.OPT_SMALL_BATCH, 0)      # bit-for-bit comparisons across block sizes: panel kernels everywhere,
    eng.set_option(capi.OPT_SPLIT, 0)            # one workgroup per tile
    eng.load_model(in_chw, layers, params, imgs.shape[0])
    out = eng.forward_host(imgs)
    eng.close()
    return out


def test_group_over_all_visible_devices_equals_single_context():
    import torch
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=9)
    imgs = synth.make_images(2 * 128 + 37, in_chw, seed=10)          # three panels on one GPU, ragged blocks on several
    want_prob, want_top5 = _single(in_chw, layers, params, imgs)
    grp = engine.QcnnDeviceGroup()
    assert grp.size == torch.cuda.device_count()
    grp.set_option(capi.OPT_KEEP_ALL, 0)
    grp.set_option(capi.OPT_SMALL_BATCH, 0)
    grp.set_option(capi.OPT_SPLIT, 0)
    grp.load_model(in_chw, layers, params, imgs.shape[0])
    assert grp.broadcast_ms is not None and grp.broadcast_ms >= 0.0
    blocks = [grp.shard_bounds(imgs.shape[0], r) for r in range(grp.size)]
    assert blocks[0][0] == 0 and blocks[-1][1] == imgs.shape[0]
    for _ in range(2):                                                 # back-to-back batches reuse the buffers
        prob, top5 = grp.forward_host(imgs)
        assert np.array_equal(prob, want_prob) and np.array_equal(top5, want_top5)
    small = imgs[: max(1, grp.size - 1)]                               # fewer images than ranks: some ranks idle
    p2, t2 = grp.forward_host(small)
    assert np.array_equal(p2, want_prob[: small.shape[0]]) and np.array_equal(t2, want_top5[: small.shape[0]])
    grp.close()


def test_two_ranks_when_two_devices_are_visible():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: the two-rank RCCL path cannot run here")
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=11)
    imgs = synth.make_images(300, in_chw, seed=12)
    want_prob, want_top5 = _single(in_chw, layers, params, imgs)
    grp = engine.QcnnDeviceGroup([0, 1])
    grp.set_option(capi.OPT_SMALL_BATCH, 0)
    grp.set_option(capi.OPT_SPLIT, 0)
    grp.load_model(in_chw, layers, params, 300)
    prob, top5 = grp.forward_host(imgs)
    assert grp.shard_bounds(300, 1) == (150, 300)
    assert np.array_equal(prob, want_prob) and np.array_equal(top5, want_top5)
    grp.close()


def test_group_rejects_duplicate_devices(monkeypatch):
    monkeypatch.delenv("QCNN_GROUP_ALLOW_DUP", raising=False)         # the test-rig allowance is opt-in
    with pytest.raises(engine.QcnnError):
        engine.QcnnDeviceGroup([0, 0])


def test_profile_ring_never_drops_forwards():
    """Per-layer timers: more forwards than the event ring holds are all accounted for (qcnn_get_layer_total_ms)."""
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=13)
    imgs = synth.make_images(3, in_chw, seed=14)
    eng = engine.QcnnEngine(0)
    eng.set_option(capi.OPT_PROFILE, 1)
    eng.set_option(capi.OPT_KEEP_ALL, 1)
    eng.load_model(in_chw, layers, params, 3)
    for _ in range(70):
        eng.forward_host(imgs)
    tot, launches, forwards = eng.layer_total_ms()
    assert forwards == 70
    conv = [i for i, l in enumerate(layers) if l["type"] == topo.CONV]
    assert all(launches[i] == 70 for i in conv) and all(tot[i] > 0 for i in conv)
    eng.close()
