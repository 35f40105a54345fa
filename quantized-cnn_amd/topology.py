"""Layer tables of the networks the reference knows (src/CaffePara.cc:20-237) and the
feature-map size rule (src/CaffeEva.cc:357-391), restated as data for the Python-side drivers
(bench.py, tests).  The C++ host library carries the same tables in CaffePara (host/caffe_para.cc).

A layer is a dict with key ``type`` in {conv, pool, fcnt, relu, lorn, drpt, smax} plus the fields
of the reference's ``LayerInfo`` that the type uses (include/CaffePara.h:28-44).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

CONV, POOL, FCNT, RELU, LORN, DRPT, SMAX = range(7)   # same order as ENUM_LyrType
TYPE_NAMES = ["conv", "pool", "fcnt", "relu", "lorn", "drpt", "smax"]


def conv(pad, knl, cnt, grp, stride):
    return dict(type=CONV, pad=pad, knl=knl, cnt=cnt, grp=grp, stride=stride)


def pool(pad, knl, stride):
    return dict(type=POOL, pad=pad, knl=knl, stride=stride)


def fcnt(nod):
    return dict(type=FCNT, nod=nod)


def relu():
    return dict(type=RELU)


def lorn(siz, alp, bet, ini):
    return dict(type=LORN, siz=siz, alp=alp, bet=bet, ini=ini)


def drpt(rat):
    return dict(type=DRPT, rat=rat)


def smax():
    return dict(type=SMAX)


def _caffenet_like(order_lrn_first: bool, drp: float, classes: int):
    l: List[dict] = [conv(0, 11, 96, 1, 4), relu()]
    if order_lrn_first:
        l += [lorn(5, 0.0001, 0.75, 1.0), pool(0, 3, 2)]
    else:
        l += [pool(0, 3, 2), lorn(5, 0.0001, 0.75, 1.0)]
    l += [conv(2, 5, 256, 2, 1), relu()]
    if order_lrn_first:
        l += [lorn(5, 0.0001, 0.75, 1.0), pool(0, 3, 2)]
    else:
        l += [pool(0, 3, 2), lorn(5, 0.0001, 0.75, 1.0)]
    l += [conv(1, 3, 384, 1, 1), relu(), conv(1, 3, 384, 2, 1), relu(),
          conv(1, 3, 256, 2, 1), relu(), pool(0, 3, 2),
          fcnt(4096), relu(), drpt(drp), fcnt(4096), relu(), drpt(drp), fcnt(classes), smax()]
    return l


def _vgg16():
    l: List[dict] = []
    for reps, ch in ((2, 64), (2, 128), (3, 256), (3, 512), (3, 512)):
        for _ in range(reps):
            l += [conv(1, 3, ch, 1, 1), relu()]
        l += [pool(0, 2, 2)]
    l += [fcnt(4096), relu(), drpt(0.5), fcnt(4096), relu(), drpt(0.5), fcnt(1000), smax()]
    return l


def _vggcnns():
    return [conv(0, 7, 96, 1, 2), relu(), lorn(5, 0.0005, 0.75, 2.0), pool(0, 3, 3),
            conv(1, 5, 256, 1, 1), relu(), pool(0, 2, 2),
            conv(1, 3, 512, 1, 1), relu(), conv(1, 3, 512, 1, 1), relu(),
            conv(1, 3, 512, 1, 1), relu(), pool(0, 3, 3),
            fcnt(4096), relu(), drpt(0.5), fcnt(4096), relu(), drpt(0.5), fcnt(1000), smax()]


# name -> (input (C,H,W), layers, default dir, default file prefix)
MODELS: Dict[str, Tuple[Tuple[int, int, int], List[dict], str, str]] = {
    "AlexNet": ((3, 227, 227), _caffenet_like(True, 0.5, 1000), "AlexNet/Bin.Files", "bvlc_alexnet_aCaF"),
    "CaffeNet": ((3, 227, 227), _caffenet_like(False, 0.5, 1000), "CaffeNet/Bin.Files", "bvlc_caffenet_aCaF"),
    "VggCnnS": ((3, 224, 224), _vggcnns(), "VggCnnS/Bin.Files", "vgg_cnn_s_aCaF"),
    "VGG16": ((3, 224, 224), _vgg16(), "VGG16/Bin.Files", "vgg16_aCaF"),
    "CaffeNetFGB": ((3, 227, 227), _caffenet_like(False, 0.7, 518), "CaffeNetFGB/Bin.Files", "bvlc_caffenetfgb_aCaF"),
    "CaffeNetFGD": ((3, 227, 227), _caffenet_like(False, 0.5, 200), "CaffeNetFGD/Bin.Files", "bvlc_caffenetfgd_aCaF"),
}


def fmap_sizes(in_chw, layers) -> List[Tuple[int, int, int]]:
    """(H, W, C) of fm[0..L]; conv floor-mode, pool ceil-mode (src/CaffeEva.cc:357-391)."""
    c, h, w = in_chw
    out = [(h, w, c)]
    for ly in layers:
        t = ly["type"]
        if t == CONV:
            h = (h + 2 * ly["pad"] - ly["knl"]) // ly["stride"] + 1
            w = (w + 2 * ly["pad"] - ly["knl"]) // ly["stride"] + 1
            c = ly["cnt"]
        elif t == POOL:
            h = int(math.ceil((h + 2 * ly["pad"] - ly["knl"]) / float(ly["stride"]))) + 1
            w = int(math.ceil((w + 2 * ly["pad"] - ly["knl"]) / float(ly["stride"]))) + 1
        elif t == FCNT:
            h, w, c = 1, 1, ly["nod"]
        out.append((h, w, c))
    return out


def tiny_model():
    """A small all-layer-types network used by the fast parity tests (not a reference model):
    exercises stride, padding, groups, CsEff < Cs, ceil-mode pooling, LRN, FC flatten, softmax."""
    layers = [conv(0, 5, 32, 1, 2), relu(), lorn(5, 0.0001, 0.75, 1.0), pool(0, 3, 2),
              conv(1, 3, 32, 2, 1), relu(), pool(0, 3, 2),
              fcnt(64), relu(), drpt(0.5), fcnt(24), smax()]
    return (3, 31, 31), layers
