"""The device-free classes of the C++ host mirror (Matrix, FileIO, CaffePara, BmpImgIO) under AddressSanitizer and
UndefinedBehaviorSanitizer (the reference has neither, and carries latent defects in these classes)."""
import os
import shutil
import subprocess

import pytest

import pyoracle as po
from conftest import ROOT, pkg

topo = pkg("topology")
synth = pkg("synth")
HOST = os.path.join(ROOT, "quantized-cnn_amd", "host")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_mirror_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(HOST, f) for f in ("host_capi.cc", "caffe_para.cc", "bmp_img_io.cc")]
    cmd = ["g++", "-std=c++11", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-DQH_NO_DEVICE", "-I" + os.path.join(ROOT, "include"),
           "-o", exe, os.path.join(ROOT, "tests", "host_sanitize_main.cc")] + srcs
    subprocess.check_call(cmd)
    in_chw, layers, _, _ = topo.MODELS["AlexNet"]
    spec = synth.quant_spec(in_chw, layers)
    params = synth.make_params(in_chw, layers, seed=5, spec=spec)
    d = str(tmp_path / "params")
    synth.write_param_dir(d, "p", params)
    args = [exe, d, "p"]
    bmp = os.path.join(po.REF_DATA, "Bmp.Files/ILSVRC2012_val_00000002.BMP")
    mean = os.path.join(po.REF_DATA, "AlexNet/imagenet_mean.single.bin")
    if os.path.exists(bmp) and os.path.exists(mean):
        args += [mean, bmp]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "host mirror under ASan/UBSan: OK" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
