"""world_size-2 gloo test of the multi-GPU host logic (quantized-cnn_amd/dist.py): rank 0 owns the parameter
values and broadcasts them, images are sharded, results gathered — and everything equals the
single-process result.  The per-rank forward pass is played by the CPU oracle here (the HIP path needs a
GPU); on the GPU box bench.py runs the same choreography with RCCL and the device arena."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg

dist_mod = pkg("dist")
topo = pkg("topology")
synth = pkg("synth")


def test_shard_bounds_cover_everything():
    for n in (1, 5, 125, 1000, 1001):
        for w in (1, 2, 3, 4, 8):
            blocks = [dist_mod.shard_bounds(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert dist_mod.shard_bounds(1000, 3, 8) == (375, 500)      # blocks of 125


def test_param_blob_roundtrip():
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=2)
    blob = dist_mod.pack_param_blob(params)
    shapes = {i: {k: p[k].shape for k in ("bias", "ctrd", "asmt")} for i, p in params.items()}
    back = dist_mod.unpack_param_blob(blob, shapes)
    for i in params:
        for k in ("bias", "ctrd", "asmt"):
            assert np.array_equal(back[i][k], params[i][k])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import torch.distributed as dist
    import pyoracle as po
    d = importlib.import_module("quantized-cnn_amd.dist")
    t = importlib.import_module("quantized-cnn_amd.topology")
    s = importlib.import_module("quantized-cnn_amd.synth")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        in_chw, layers = t.tiny_model()
        spec = s.quant_spec(in_chw, layers, conv_k=128)
        full = s.make_params(in_chw, layers, seed=2)          # only rank 0 is allowed to USE these values
        shapes = {i: {k: full[i][k].shape for k in ("bias", "ctrd", "asmt")} for i in full}
        params = d.broadcast_params(full if rank == 0 else None, shapes, src=0)
        n = 11
        imgs = s.make_images(n, in_chw, seed=6)               # synthetic batch, same seed everywhere
        lo, hi = d.shard_bounds(n, rank, world)
        orc = po.COracle(in_chw, layers)
        orc.set_params(params)
        orc.forward(imgs[lo:hi])
        prob = orc.fm(len(layers)).reshape(hi - lo, -1)
        top5 = np.stack([orc.top5(p) for p in prob]).astype(np.int32)
        all_prob = d.gather_rows(prob, n)
        all_top5 = d.gather_rows(top5, n)
        dist.barrier()
        q.put((rank, all_prob, all_top5, params[0]["ctrd"].sum()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_single_process():
    import torch.multiprocessing as mp
    import pyoracle as po
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    in_chw, layers = topo.tiny_model()
    params = synth.make_params(in_chw, layers, seed=2)
    imgs = synth.make_images(11, in_chw, seed=6)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs)
    want = orc.fm(len(layers)).reshape(11, -1)
    for rank, prob, top5, csum in res:
        assert np.array_equal(prob, want), "rank %d" % rank
        assert np.array_equal(top5, np.stack([orc.top5(p) for p in want]))
        assert csum == params[0]["ctrd"].sum()


def _checksum_worker(rank, world, port, q, corrupt):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    d = importlib.import_module("quantized-cnn_amd.dist")
    s = importlib.import_module("quantized-cnn_amd.synth")
    t = importlib.import_module("quantized-cnn_amd.topology")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        in_chw, layers = t.tiny_model()
        full = s.make_params(in_chw, layers, seed=2)
        shapes = {i: {k: full[i][k].shape for k in ("bias", "ctrd", "asmt")} for i in full}
        params = d.broadcast_params(full if rank == 0 else None, shapes, src=0)
        blob = d.pack_param_blob(params)
        if corrupt and rank == 1:
            blob[blob.size // 2] ^= 1                          # one flipped bit on one rank
        w = np.frombuffer(blob.tobytes() + b"\0" * (-blob.size % 4), dtype="<u4").astype(np.uint64)
        idx = (np.arange(w.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
        pair = (int(w.sum(dtype=np.uint64)), int((w * idx).sum(dtype=np.uint64)))      # the sums k_arena_checksum forms, modulo 2^64
        ok, pairs = d.checksums_agree(pair)
        q.put((rank, ok, pairs, d.verified_world_size()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_broadcast_checksums_across_ranks(corrupt):
    """What bench.py --gpus N does between dist.broadcast(arena) and the timed loop (and qcnn_group_model_broadcast does inside
    one process): every rank forms the checksum pair of ITS copy of the parameters, all ranks compare — equal after a sound
    broadcast, unequal (on every rank) when one rank's copy differs in one bit; the world size is the one a real all_reduce counts."""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_checksum_worker, args=(r, 2, port, q, corrupt)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, pairs, world in res:
        assert world == 2 and len(pairs) == 2
        assert ok == (not corrupt), "rank %d: %r" % (rank, pairs)
        assert (pairs[0] == pairs[1]) == (not corrupt)
        assert all(0 <= v < 2 ** 64 for p in pairs for v in p)
    assert res[0][2] == res[1][2]                               # every rank sees the same table


def test_bench_refuses_a_multi_gpu_run_it_cannot_do():
    """`python bench.py --gpus 2` outside torch.distributed.run must either start two ranks or fail loudly — never
    report a one-GPU number as a two-GPU one (there is no GPU in the CPU tier, so it has to fail)."""
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the run would be legitimate")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stdout + r.stderr)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())
