"""File formats of the hot path's parameters (.bin / .cbn) — host logic, CPU only."""
import os

import numpy as np
import pytest

from conftest import pkg

fileio = pkg("fileio")


def test_bin_roundtrip(tmp_path):
    for shape, dt in (((7,), np.float32), ((3, 5), np.uint16), ((2, 3, 4), np.float32), ((2, 3, 4, 5), np.uint8)):
        a = (np.arange(np.prod(shape)) % 251).astype(dt).reshape(shape)
        p = str(tmp_path / "a.bin")
        fileio.write_bin(p, a)
        assert os.path.getsize(p) == 4 + 4 * len(shape) + a.nbytes
        b = fileio.read_bin(p, dt)
        assert b.shape == a.shape and np.array_equal(a, b)


@pytest.mark.parametrize("bits,shape", [(7, (96, 11, 11, 1)), (5, (64, 72)), (4, (24, 64)), (7, (3, 5)), (1, (9,)),
                                        (8, (5000,)), (5, (6553,)), (5, (6554,)), (7, (4681,)), (7, (4682,))])
def test_cbn_roundtrip_and_size(tmp_path, bits, shape):
    rng = np.random.default_rng(bits * 1000 + len(shape))
    a = rng.integers(0, 1 << bits, size=shape, dtype=np.uint8)
    p = str(tmp_path / "a.cbn")
    fileio.write_cbn(p, a, bits)
    per = (4096 * 8) // bits
    nblk = (a.size + per - 1) // per
    assert os.path.getsize(p) == 4 + 4 * len(shape) + 4 + nblk * 4096      # header + whole blocks
    b, bits2 = fileio.read_cbn(p)
    assert bits2 == bits and b.shape == a.shape and np.array_equal(a, b)


def test_cbn_known_bytes(tmp_path):
    """Hand-computed vector: three 5-bit values 10101 00011 11111 -> bit stream 10101000 1111111(0) ->
    bytes A8 FE 00 (MSB first, include/FileIO.h:299-341)."""
    p = str(tmp_path / "k.cbn")
    fileio.write_cbn(p, np.array([21, 3, 31], np.uint8), 5)
    raw = open(p, "rb").read()
    assert raw[:12] == b"\x01\x00\x00\x00\x03\x00\x00\x00\x05\x00\x00\x00"
    assert raw[12:15] == bytes([0xA8, 0xFE, 0x00])
    assert len(raw) == 12 + 4096


def test_fc6_blob_size(tmp_path):
    """the synthesised fc6 assignment file is 5 902 352 bytes."""
    per = (4096 * 8) // 5
    n = 4096 * 2304
    assert 16 + ((n + per - 1) // per) * 4096 == 5902352


def test_cbn_matches_c_oracle_decoder(tmp_path):
    import pyoracle as po
    topo = pkg("topology")
    in_chw, layers = topo.tiny_model()
    orc = po.COracle(in_chw, layers)
    for bits in (4, 5, 7):
        a = np.random.default_rng(bits).integers(0, 1 << bits, size=(10007,), dtype=np.uint8)
        p = str(tmp_path / "x.cbn")
        fileio.write_cbn(p, a, bits)
        raw = np.frombuffer(open(p, "rb").read()[12:], np.uint8).copy()
        out = np.zeros(a.size, np.uint8)
        orc.lib.qo_cbn_decode(raw, a.size, bits, out)
        assert np.array_equal(out, a)


def test_min_bits():
    assert fileio.min_bits(np.array([127])) == 7
    assert fileio.min_bits(np.array([31])) == 5
    assert fileio.min_bits(np.array([15])) == 4
    assert fileio.min_bits(np.array([128])) == 8
    assert fileio.min_bits(np.array([0])) == 1
