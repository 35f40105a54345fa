"""Build recipes: hipcc (gfx950) for the device library, g++ for the C++ host mirror.

Everything is built IN-TREE so that the artefacts travel to the GPU box with the repo snapshot:
  quantized-cnn_amd/libqcnn_hip.so    HIP kernels + C-ABI (include/qcnn_hip.h)
  quantized-cnn_amd/libqcnn_host.so   C++ host mirror of the reference interface (CaffeEva, CaffePara, ...)
  build/bin/qcnn_main                 this repo's own driver over the host mirror
  build/bin/QuanCNN_hip               the reference's UNMODIFIED Main.cc / UnitTest.cc linked against
                                      the host mirror (only where /root/reference exists; staged copies,
                                      never committed)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
HIP_SO = os.path.join(PKG, "libqcnn_hip.so")
HOST_SO = os.path.join(PKG, "libqcnn_host.so")
BIN_DIR = os.path.join(ROOT, "build", "bin")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

HIP_SOURCES = ["qcnn_kernels.hip", "qcnn_sym8.hip", "qcnn_half8.hip", "qcnn_planner.hip", "qcnn_glue.hip", "qcnn_small.hip", "qcnn_dense.hip", "qcnn_decoded.hip", "qcnn_engine.hip", "qcnn_group.hip"]
HIP_FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-ffp-contract=off", "-Wall",
             "-Wno-unused-function"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd, **kw):
    print("[build] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, **kw)


def build_hip(force: bool = False, extra_flags=()) -> str:
    """Compile the HIP extension for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "qcnn_kernels.h"), os.path.join(CSRC, "qcnn_dev.h"), os.path.join(CSRC, "qcnn_sym8_gather.h"), os.path.join(CSRC, "qcnn_planner.h"), os.path.join(ROOT, "include", "qcnn_hip.h")]
    if force or _newer(HIP_SO, deps):
        objs = []
        for s in srcs:
            o = os.path.join(CSRC, os.path.basename(s) + ".o")
            if force or _newer(o, deps):
                _run([HIPCC] + HIP_FLAGS + list(extra_flags) + ["-c", s, "-o", o])
            objs.append(o)
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", HIP_SO] + objs +
             ["-L/opt/rocm/lib", "-lrccl", "-lpthread"])     # RCCL: the device group's parameter broadcast
    return HIP_SO


def build_host(force: bool = False):
    """Compile the C++ host mirror (plain g++, links libqcnn_hip.so) and this repo's driver."""
    if not os.path.isdir(HOST):
        return None
    srcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".cc") and f != "qcnn_main.cc")
    if not srcs:
        return None
    incs = [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if force or _newer(HOST_SO, srcs + incs):
        _run(["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(ROOT, "include"),
              "-o", HOST_SO] + srcs + ["-L" + PKG, "-lqcnn_hip", "-Wl,-rpath," + PKG])
    main_cc = os.path.join(HOST, "qcnn_main.cc")
    exe = os.path.join(BIN_DIR, "qcnn_main")
    if os.path.exists(main_cc) and (force or _newer(exe, [main_cc, HOST_SO])):
        os.makedirs(BIN_DIR, exist_ok=True)
        _run(["g++", "-std=c++11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe, main_cc,
              "-L" + PKG, "-lqcnn_host", "-lqcnn_hip", "-Wl,-rpath," + PKG])
    return HOST_SO


def build_reference_driver(ref: str = "/root/reference", force: bool = False):
    """Drive the reference's byte-identical Main.cc + UnitTest.cc against the host mirror
    (SURVEY.md §0 fact 9): stage copies under build/stage/src beside build/stage/include -> include/."""
    if not os.path.isdir(os.path.join(ref, "src")) or not os.path.exists(HOST_SO):
        return None
    if not os.path.exists(os.path.join(HOST, "caffe_eva_wrapper.cc")):
        return None     # host mirror incomplete: nothing to link the reference drivers against
    stage = os.path.join(ROOT, "build", "stage")
    os.makedirs(os.path.join(stage, "src"), exist_ok=True)
    inc_link = os.path.join(stage, "include")
    if not os.path.islink(inc_link):
        if os.path.exists(inc_link):
            shutil.rmtree(inc_link)
        os.symlink(os.path.join(ROOT, "include"), inc_link)
    for f in ("Main.cc", "UnitTest.cc"):
        shutil.copyfile(os.path.join(ref, "src", f), os.path.join(stage, "src", f))
    exe = os.path.join(BIN_DIR, "QuanCNN_hip")
    srcs = [os.path.join(stage, "src", f) for f in ("Main.cc", "UnitTest.cc")]
    if force or _newer(exe, srcs + [HOST_SO]):
        os.makedirs(BIN_DIR, exist_ok=True)
        _run(["g++", "-std=c++11", "-O2", "-w", "-o", exe] + srcs +
             ["-L" + PKG, "-lqcnn_host", "-lqcnn_hip", "-Wl,-rpath," + PKG])
    return exe


def build_planner_cpu(force: bool = False) -> str:
    """The launch planner (csrc/qcnn_planner.hip: host code only) compiled by g++ into build/libqcnn_planner_cpu.so — the CPU
    test tier exercises the very functions libqcnn_hip.so plans with (tests/test_planner_cpu.py), no GPU, no hipcc."""
    out = os.path.join(ROOT, "build", "libqcnn_planner_cpu.so")
    src = os.path.join(CSRC, "qcnn_planner.hip")
    deps = [src, os.path.join(CSRC, "qcnn_planner.h"), os.path.join(CSRC, "qcnn_kernels.h")]
    if force or _newer(out, deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run(["g++", "-std=c++17", "-O2", "-x", "c++", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
              "-I" + CSRC, src, "-o", out])
    return out


def build_all(force: bool = False):
    build_hip(force)
    build_host(force)
    build_reference_driver(force=force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
