"""pytest wiring: `gpu` marker, import paths, shared fixtures.

CPU tier  (`-m "not gpu"`): oracle vs golden vectors / compiled reference, file formats, host logic,
                            C-ABI symbol check, world_size-2 gloo sharding test.
GPU tier  (`-m gpu`):       HIP path (through the C-ABI) vs the oracle and the golden vectors.
"""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pkg(name=""):
    return importlib.import_module("quantized-cnn_amd" + ("." + name if name else ""))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_tiny():
    return np.load(os.path.join(GOLDEN, "tiny_ref.npz"))


@pytest.fixture(scope="session")
def golden_alex_real():
    return np.load(os.path.join(GOLDEN, "alexnet_real_ref.npz"))


@pytest.fixture(scope="session")
def golden_alex_syn():
    return np.load(os.path.join(GOLDEN, "alexnet_syn_ref.npz"))


def tiny_params_from_golden(z, layers):
    topo = pkg("topology")
    params = {}
    for i, ly in enumerate(layers):
        if ly["type"] in (topo.CONV, topo.FCNT):
            params[i] = dict(bias=z["bias_%02d" % i], ctrd=z["ctrd_%02d" % i], asmt=z["asmt_%02d" % i], bits=7)
    return params


def fingerprint(a):
    d = np.asarray(a, np.float64).reshape(-1)
    return np.array([d.sum(), np.abs(d).sum(), np.sqrt((d * d).sum()), d.min(), d.max()], np.float64)


def rel_err(a, b):
    """max-norm and l2 relative error of a against reference b (SURVEY.md §7 'Accumulation order')."""
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    den_inf = max(np.abs(b).max(), 1e-30)
    den_l2 = max(np.sqrt((b * b).sum()), 1e-30)
    return np.abs(a - b).max() / den_inf, np.sqrt(((a - b) ** 2).sum()) / den_l2


@pytest.fixture(scope="session")
def golden_alex_real10():
    return np.load(os.path.join(GOLDEN, "alexnet_real10_ref.npz"))


def real_bmp_images():
    """The ten shipped Bmp.Files/*.BMP as the network input [10, 3, 227, 227] (mean-subtracted BGR crop): through the
    compiled reference's own BmpImgIO where oracle/_ref/libqcnn_ref.so exists, else through the host mirror's (which
    tests/test_host_mirror.py holds bit-identical to it).  Needs oracle/_ref/data."""
    import ctypes as C
    import pyoracle as po
    mean = os.path.join(po.REF_DATA, "AlexNet/imagenet_mean.single.bin")
    bmps = [os.path.join(po.REF_DATA, "Bmp.Files/ILSVRC2012_val_%08d.BMP" % i) for i in range(1, 11)]
    if po.have_ref():
        ref = po.RefLib()
        return np.concatenate([ref.load_bmp(mean, b) for b in bmps])
    lib = C.CDLL(os.path.join(ROOT, "quantized-cnn_amd", "libqcnn_host.so"))
    lib.qh_bmp_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")]
    out = np.empty((10, 3, 227, 227), np.float32)
    for i, b in enumerate(bmps):
        with po._Quiet():
            assert lib.qh_bmp_load(mean.encode(), b.encode(), 256, 227, 0, out[i:i + 1]) == 0
    return out
