"""Q-CNN parameter / dataset file formats, numpy side.

Two on-disk formats carry every tensor of the hot path (SURVEY.md §8b "Data contracts"):

``.bin``  little-endian ``int32 dimCnt; int32 dims[dimCnt]; T data[]`` row-major.
          Reference reader: include/FileIO.h:55-107, writer :240-297.
``.cbn``  ``int32 dimCnt; int32 dims[]; int32 bitCntPerEle;`` followed by 4096-byte blocks.
          Each block holds ``floor(32768 / bits)`` indices packed MSB-first as one contiguous
          bit stream; an index never straddles a block; the stored value is the 0-based index
          (the reference reader adds 1, include/FileIO.h:165, and CaffePara::LoadLayerPara
          subtracts it again, src/CaffePara.cc:285-288).  The file is padded to whole blocks.
          Reference reader: include/FileIO.h:109-178, writer :299-350.

This module is host-side plumbing used by bench.py / tests to synthesise parameter sets in the
reference's own file layout; the C++ host library has its own reader (include/FileIO.h of this repo).
"""
from __future__ import annotations

import os
import struct

import numpy as np

CBN_BLOCK_BYTES = 4096
_DTYPES = {
    "single": np.float32,
    "float32": np.float32,
    "uint8": np.uint8,
    "uint16": np.uint16,
    "int32": np.int32,
}


def write_bin(path: str, arr: np.ndarray) -> None:
    """Write ``arr`` (1..4-D) as a reference ``.bin`` file."""
    arr = np.ascontiguousarray(arr)
    if not 1 <= arr.ndim <= 4:
        raise ValueError("Matrix<T> supports 1..4 dimensions, got %d" % arr.ndim)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", arr.ndim))
        f.write(struct.pack("<%di" % arr.ndim, *arr.shape))
        f.write(arr.tobytes())


def read_bin(path: str, dtype) -> np.ndarray:
    """Read a reference ``.bin`` file; ``dtype`` is the element type (not stored in the file)."""
    dtype = np.dtype(_DTYPES.get(dtype, dtype))
    with open(path, "rb") as f:
        (ndim,) = struct.unpack("<i", f.read(4))
        if not 1 <= ndim <= 4:
            raise ValueError("%s: bad dimCnt %d" % (path, ndim))
        dims = struct.unpack("<%di" % ndim, f.read(4 * ndim))
        n = int(np.prod(dims))
        data = np.fromfile(f, dtype=dtype, count=n)
    if data.size != n:
        raise ValueError("%s: truncated (%d of %d elements)" % (path, data.size, n))
    return data.reshape(dims)


def cbn_vals_per_block(bits: int) -> int:
    return (CBN_BLOCK_BYTES * 8) // bits


def cbn_pack(idx0: np.ndarray, bits: int) -> np.ndarray:
    """The payload of a ``.cbn`` file (everything behind the header) for 0-based indices ``idx0``: uint8 array."""
    flat = np.ascontiguousarray(idx0, dtype=np.uint8).reshape(-1)
    if flat.size and int(flat.max()) >= (1 << bits):
        raise ValueError("index %d does not fit %d bits" % (int(flat.max()), bits))
    per = cbn_vals_per_block(bits)
    nblk = (flat.size + per - 1) // per
    shifts = np.arange(bits - 1, -1, -1, dtype=np.uint8)
    out = np.zeros(nblk * CBN_BLOCK_BYTES, np.uint8)
    for b in range(nblk):
        v = flat[b * per:(b + 1) * per]
        bitmat = ((v[:, None] >> shifts[None, :]) & 1).astype(np.uint8).reshape(-1)
        blk = np.zeros(CBN_BLOCK_BYTES * 8, dtype=np.uint8)
        blk[:bitmat.size] = bitmat
        out[b * CBN_BLOCK_BYTES:(b + 1) * CBN_BLOCK_BYTES] = np.packbits(blk)
    return out


def write_cbn(path: str, idx0: np.ndarray, bits: int) -> None:
    """Write 0-based indices ``idx0`` (uint8, 1..4-D) bit-packed with ``bits`` bits per element."""
    idx0 = np.ascontiguousarray(idx0, dtype=np.uint8)
    if not 1 <= bits <= 8:
        raise ValueError("bitCntPerEle must be 1..8")
    if idx0.size and int(idx0.max()) >= (1 << bits):
        raise ValueError("index %d does not fit %d bits" % (int(idx0.max()), bits))
    flat = idx0.reshape(-1)
    per = cbn_vals_per_block(bits)
    nblk = (flat.size + per - 1) // per
    shifts = np.arange(bits - 1, -1, -1, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", idx0.ndim))
        f.write(struct.pack("<%di" % idx0.ndim, *idx0.shape))
        f.write(struct.pack("<i", bits))
        for b in range(nblk):
            v = flat[b * per:(b + 1) * per]
            bitmat = ((v[:, None] >> shifts[None, :]) & 1).astype(np.uint8).reshape(-1)
            blk = np.zeros(CBN_BLOCK_BYTES * 8, dtype=np.uint8)
            blk[:bitmat.size] = bitmat
            f.write(np.packbits(blk).tobytes())


def read_cbn(path: str):
    """Read a ``.cbn`` file.  Returns ``(idx0 uint8 array, bits)`` with 0-based indices."""
    with open(path, "rb") as f:
        (ndim,) = struct.unpack("<i", f.read(4))
        dims = struct.unpack("<%di" % ndim, f.read(4 * ndim))
        (bits,) = struct.unpack("<i", f.read(4))
        n = int(np.prod(dims))
        per = cbn_vals_per_block(bits)
        nblk = (n + per - 1) // per
        raw = np.frombuffer(f.read(nblk * CBN_BLOCK_BYTES), dtype=np.uint8)
    if raw.size != nblk * CBN_BLOCK_BYTES:
        raise ValueError("%s: truncated" % path)
    out = np.empty(nblk * per, dtype=np.uint8)
    weights = (1 << np.arange(bits - 1, -1, -1)).astype(np.uint16)
    for b in range(nblk):
        blkbits = np.unpackbits(raw[b * CBN_BLOCK_BYTES:(b + 1) * CBN_BLOCK_BYTES])
        vals = blkbits[:per * bits].reshape(per, bits).astype(np.uint16) @ weights
        out[b * per:(b + 1) * per] = vals.astype(np.uint8)
    return out[:n].reshape(dims), bits


def min_bits(idx0: np.ndarray) -> int:
    """Smallest bit width that holds every 0-based index (CaffePara::CalcBitCntPerEle,
    src/CaffePara.cc:360-380, restated for 0-based input)."""
    m = int(idx0.max()) if idx0.size else 0
    bits = 0
    while m:
        m >>= 1
        bits += 1
    return max(bits, 1)


def param_path(dir_path: str, prefix: str, kind: str, layer_1based: int, ext: str) -> str:
    """``<dir>/<pfx>.<kind>.<NN>.<ext>`` with NN the 1-based layer index (src/CaffePara.cc:262-281)."""
    return os.path.join(dir_path, "%s.%s.%02d.%s" % (prefix, kind, layer_1based, ext))
