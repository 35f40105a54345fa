#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — stage the reference's shipped DATA (not sources) into oracle/_ref/data.

oracle/_ref/ is git-ignored but not gpurun-ignored, so what is staged here travels to the GPU box
next to the prebuilt oracle/_ref/libqcnn_ref.so, while the history stays free of reference files.
Staged: AlexNet/Bin.Files/* (23 of 24 parameter files), AlexNet/imagenet_mean.single.bin,
Bmp.Files/*.BMP, Cls.Names/*, ILSVRC12.227x227.IMG/lablVecTst.uint16.bin, plus the one blob the
mount lacks, synthesised with the SURVEY.md §8c recipe:

  AlexNet/Bin.Files/bvlc_alexnet_aCaF.asmtLst.16.cbn   fc6 assignments, dims (4096, 2304), 5 bits,
      numpy default_rng(0).integers(0, 32)  -> 5 902 352 bytes                       ("fixture 1")
  AlexNet/fixtures/bvlc_alexnet_aCaF.asmtLst.16.fx2.cbn   a second fc6 table with which fc7 / fc8 / top-5 of
      the real network are NOT degenerate (synth.fc6_fixture(ctrd, 2): small-norm code words) ("fixture 2")

Run by ``__graft_entry__.build()`` when /root/reference exists.  Usage: stage_ref_data.py [REF]
"""
import importlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
fileio = importlib.import_module("quantized-cnn_amd.fileio")
synth = importlib.import_module("quantized-cnn_amd.synth")

COPY = ["AlexNet/Bin.Files", "AlexNet/imagenet_mean.single.bin", "Bmp.Files", "Cls.Names",
        "ILSVRC12.227x227.IMG/lablVecTst.uint16.bin"]


def stage(ref: str = "/root/reference", out: str = os.path.join(HERE, "_ref", "data")) -> str:
    for rel in COPY:
        src, dst = os.path.join(ref, rel), os.path.join(out, rel)
        if os.path.isdir(src):
            os.makedirs(dst, exist_ok=True)
            for name in sorted(os.listdir(src)):
                if not os.path.exists(os.path.join(dst, name)):
                    shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
        else:
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            if not os.path.exists(dst):
                shutil.copyfile(src, dst)
    fc6 = os.path.join(out, "AlexNet/Bin.Files/bvlc_alexnet_aCaF.asmtLst.16.cbn")
    if not os.path.exists(fc6):
        idx = np.random.default_rng(0).integers(0, 32, size=(4096, 2304), dtype=np.uint8)
        fileio.write_cbn(fc6, idx, 5)
    fx2 = os.path.join(out, synth.FC6_FIXTURE2_NAME)
    if not os.path.exists(fx2):
        os.makedirs(os.path.dirname(fx2), exist_ok=True)
        ctrd = fileio.read_bin(os.path.join(out, "AlexNet/Bin.Files/bvlc_alexnet_aCaF.ctrdLst.16.bin"), np.float32)
        fileio.write_cbn(fx2, synth.fc6_fixture(ctrd, 2), 5)
    return out


if __name__ == "__main__":
    print(stage(*(sys.argv[1:2])))
