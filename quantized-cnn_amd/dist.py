"""Multi-GPU host logic: image sharding and the one-time parameter broadcast.

The forward pass has no data-path collective: images are independent, every GPU holds the full
parameter set.  One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests):

  * rank 0 owns the parameter VALUES; ``broadcast_params`` ships them to the other ranks as one flat
    byte blob (biases, codebooks, assignments in the reference's file layout).  bench.py uses the
    device-side variant of the same idea: it broadcasts rank 0's packed device arena
    (qcnn_model_commit(dev_arena) + dist.broadcast + qcnn_model_mark_loaded).
  * ``shard_bounds`` gives rank r the contiguous block [r*N/W, (r+1)*N/W) of a global batch.
  * ``gather_rows`` reassembles per-rank results (top-5 / probabilities) on every rank.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rank `rank`: item i goes to rank i*world // n_items (blocks differ by <= 1)."""
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


def blob_layout(shapes: Dict[int, dict]):
    """shapes: {layer: dict(bias=(Ct,), ctrd=(M,K,Cs), asmt=(...))}.  Returns [(layer, key, dtype, shape, off)], total."""
    items, off = [], 0
    for i in sorted(shapes):
        for key, dt in (("bias", np.float32), ("ctrd", np.float32), ("asmt", np.uint8)):
            shp = tuple(int(x) for x in shapes[i][key])
            nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
            items.append((i, key, dt, shp, off))
            off = (off + nbytes + 15) // 16 * 16
    return items, off


def pack_param_blob(params) -> np.ndarray:
    shapes = {i: {k: p[k].shape for k in ("bias", "ctrd", "asmt")} for i, p in params.items()}
    items, total = blob_layout(shapes)
    blob = np.zeros(total, np.uint8)
    for i, key, dt, shp, off in items:
        raw = np.ascontiguousarray(params[i][key], dt).view(np.uint8).reshape(-1)
        blob[off:off + raw.size] = raw
    return blob


def unpack_param_blob(blob: np.ndarray, shapes):
    items, total = blob_layout(shapes)
    assert blob.size == total, "parameter blob size mismatch"
    out: Dict[int, dict] = {}
    for i, key, dt, shp, off in items:
        n = int(np.prod(shp)) * np.dtype(dt).itemsize
        out.setdefault(i, {})[key] = blob[off:off + n].view(dt).reshape(shp).copy()
    return out


def broadcast_params(params, shapes, src: int = 0, device="cpu"):
    """Rank `src` passes its parameter dict, the others pass None; everyone returns the full dict.
    `shapes` (known to every rank from the topology + quantisation spec) fixes the blob layout."""
    import torch
    import torch.distributed as dist
    _, total = blob_layout(shapes)
    if dist.get_rank() == src:
        t = torch.from_numpy(pack_param_blob(params)).to(device)
    else:
        t = torch.empty(total, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return unpack_param_blob(t.cpu().numpy(), shapes)


def gather_rows(local_rows, n_items: int):
    """All-gather per-rank row blocks (shard_bounds order) into the full [n_items, ...] array on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    local = torch.as_tensor(np.ascontiguousarray(local_rows))
    width = tuple(local.shape[1:])
    cap = max(shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world))
    pad = torch.zeros((cap,) + width, dtype=local.dtype)
    pad[: local.shape[0]] = local
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = []
    for r in range(world):
        lo, hi = shard_bounds(n_items, r, world)
        out.append(bufs[r][: hi - lo])
    return torch.cat(out, 0).numpy()


def checksums_agree(local_pair, device="cpu"):
    """Every rank passes its (sum, weighted sum) checksum pair of the parameter arena (QcnnEngine.arena_checksum, or any pair of
    Python ints < 2^64); returns (ok, [pair of rank 0, pair of rank 1, ...]) on every rank — ok iff all ranks hold rank 0's pair.
    What bench.py --gpus N checks between dist.broadcast(arena) and the timed loop."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    # 64-bit unsigned values travel as four 16-bit limbs each in int64 slots (no unsigned 64-bit tensors in every backend)
    limbs = [(int(v) >> (16 * k)) & 0xffff for v in local_pair for k in range(4)]
    mine = torch.tensor(limbs, dtype=torch.int64, device=device)
    bufs = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    pairs = []
    for b in bufs:
        v = [int(x) for x in b.cpu().tolist()]
        pairs.append(tuple(sum(v[4 * j + k] << (16 * k) for k in range(4)) for j in range(2)))
    return all(p == pairs[0] for p in pairs), pairs


def verified_world_size(device="cpu"):
    """World size as the communicator itself reports it after a REAL collective: all_reduce of ones."""
    import torch
    import torch.distributed as dist
    t = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(t)
    n = int(t.item())
    if n != dist.get_world_size():
        raise RuntimeError("all_reduce over the communicator counted %d ranks, get_world_size() says %d" % (n, dist.get_world_size()))
    return n
