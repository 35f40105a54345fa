"""The C-ABI library loads on a CPU-only host and exports every symbol include/qcnn_hip.h declares;
compute entry points fail loudly (no CPU fallback)."""
import ctypes as C

import pytest

from conftest import has_gpu, pkg

capi = pkg("capi")


def test_exports_every_declared_symbol():
    lib = capi.load()
    names = capi.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libqcnn_hip.so does not export %s" % n
    assert lib.qcnn_abi_version() == 5


@pytest.mark.skipif(has_gpu(), reason="only meaningful without a GPU")
def test_no_silent_cpu_fallback():
    lib = capi.load()
    h = C.c_void_p()
    rc = lib.qcnn_ctx_create(0, None, C.byref(h))
    assert rc != 0 and not h
    msg = lib.qcnn_last_error(None).decode()
    assert "no HIP device" in msg or "no CPU path" in msg
    engine = pkg("engine")
    with pytest.raises(engine.QcnnError):
        engine.QcnnEngine(0)


def test_topology_matches_survey_sizes():
    topo = pkg("topology")
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    s = topo.fmap_sizes(in_chw, layers)
    assert s[0] == (227, 227, 3) and s[1] == (55, 55, 96) and s[4] == (27, 27, 96) and s[5] == (27, 27, 256)
    assert s[8] == (13, 13, 256) and s[9] == (13, 13, 384) and s[15] == (6, 6, 256) and s[16] == (1, 1, 4096)
    assert s[23] == (1, 1, 1000)
    assert sum(h * w * c for h, w, c in s) == 2080811          # sum of feature maps
    in_chw, layers, _, _ = topo.MODELS["VGG16"]
    s = topo.fmap_sizes(in_chw, layers)
    assert len(layers) == 39 and s[31] == (7, 7, 512) and s[-1] == (1, 1, 1000)
    spec = pkg("synth").quant_spec(*topo.MODELS["AlexNet"][:2])
    assert {i: (v["M"], v["K"], v["Cs"]) for i, v in spec.items()} == {
        0: (1, 128, 8), 4: (6, 128, 8), 8: (32, 128, 8), 10: (24, 128, 8), 12: (24, 128, 8),
        15: (2304, 32, 4), 18: (1024, 32, 4), 21: (4096, 16, 1)}


def test_python_option_constants_match_the_header():
    """quantized-cnn_amd/capi.py restates the QCNN_OPT_* / QCNN_SMALL_BATCH_MAX values of include/qcnn_hip.h."""
    import re
    text = open(capi.HEADER_PATH).read()
    enum = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"^\s+QCNN_OPT_([A-Z0-9_]+)\s*=\s*(\d+),?\s*/\*", text, re.M))
    assert len(enum) >= 13
    for name, value in enum.items():
        assert getattr(capi, "OPT_" + name) == value, name
    assert capi.SMALL_BATCH_MAX == int(re.search(r"#define\s+QCNN_SMALL_BATCH_MAX\s+(\d+)", text).group(1))
