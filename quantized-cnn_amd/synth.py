"""Synthetic Q-CNN parameter sets in the reference's own file layout.

The reference ships AlexNet parameters only (and its fc6 assignment file is missing from the mount,
SURVEY.md §0 fact 3); the GPU box has no /root/reference at all.  bench.py and the GPU parity tests
therefore run on parameter sets generated here from a fixed seed: same shapes, same value ranges,
same files (``<pfx>.biasVec.NN.bin``, ``.ctrdLst.NN.bin``, ``.asmtLst.NN.cbn``; src/CaffePara.cc:262-281)
as the shipped ones, so that the reference, the oracle and the HIP path all load them through their
own readers.

Quantisation layout rule (decoded from the shipped headers): conv layers use
Cs = 8 dims per subspace and K = 128 codewords, M = ceil(Cin_per_group / 8) subspaces (so the first
conv, Cin = 3, has M = 1 and uses only 3 of the 8 codebook dims, src/CaffeEva.cc:1277); hidden FC
layers use Cs = 4, K = 32; the classifier FC uses Cs = 1, K = 16.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np

from . import fileio
from .topology import CONV, FCNT, fmap_sizes


def quant_spec(in_chw, layers, conv_k=128, conv_cs=8, fc_k=32, fc_cs=4, last_k=16, last_cs=1):
    """Per conv/FC layer: dict(M, K, Cs, Ct, kind, knl, D) following the shipped layout rule."""
    sizes = fmap_sizes(in_chw, layers)
    fc_idx = [i for i, ly in enumerate(layers) if ly["type"] == FCNT]
    spec: Dict[int, dict] = {}
    for i, ly in enumerate(layers):
        h, w, c = sizes[i]
        if ly["type"] == CONV:
            cg = c // ly["grp"]
            m = (cg + conv_cs - 1) // conv_cs
            spec[i] = dict(kind="conv", M=m, K=conv_k, Cs=conv_cs, Ct=ly["cnt"], knl=ly["knl"], D=cg)
        elif ly["type"] == FCNT:
            d = h * w * c
            last = (i == fc_idx[-1]) and len(fc_idx) > 1
            k, cs = (last_k, last_cs) if last else (fc_k, fc_cs)
            if d % cs:
                raise ValueError("FC input %d not divisible by Cs=%d" % (d, cs))
            spec[i] = dict(kind="fc", M=d // cs, K=k, Cs=cs, Ct=ly["nod"], knl=1, D=d)
    return spec


def make_params(in_chw, layers, seed=0, spec: Optional[Dict[int, dict]] = None):
    """Random parameters.  Returns {layer_idx: dict(bias[Ct] f32, ctrd[M,K,Cs] f32 (file order),
    asmt uint8 0-based in file order ([Ct,kh,kw,M] conv / [Ct,M] fc), bits)}."""
    spec = spec or quant_spec(in_chw, layers)
    rng = np.random.default_rng(seed)
    out = {}
    first = min(spec) if spec else -1
    for i in sorted(spec):
        s = spec[i]
        fan_in = s["knl"] * s["knl"] * s["D"]
        # He-style scale keeps the activation variance layer to layer; the first layer also brings the
        # +-128 pixel range down to O(1) so that the reference's un-shifted softmax (expf, no max
        # subtraction, src/CaffeEva.cc:1107-1114) stays finite.
        scale = np.float32(np.sqrt(2.0 / fan_in) * (1.0 / 64.0 if i == first else 1.0))
        ctrd = (rng.standard_normal((s["M"], s["K"], s["Cs"])) * scale).astype(np.float32)
        bias = (rng.standard_normal(s["Ct"]) * 0.1).astype(np.float32)
        if s["kind"] == "conv":
            shape = (s["Ct"], s["knl"], s["knl"], s["M"])
        else:
            shape = (s["Ct"], s["M"])
        asmt = rng.integers(0, s["K"], size=shape, dtype=np.uint8)
        out[i] = dict(bias=bias, ctrd=ctrd, asmt=asmt, bits=fileio.min_bits(np.array([s["K"] - 1])))
    return out


def make_dense_params(in_chw, layers, seed=0):
    """Random parameters of the reference's PRECISE path (Init(false), src/CaffePara.cc:290-302): per conv/FC layer
    dict(bias [Ct], weights) with conv kernels [Ct][Cin/grp][kh][kw] (convKnl.NN.bin) or FC weights [Ct][D]
    (fcntWei.NN.bin).  Same scaling rule as make_params."""
    sizes = fmap_sizes(in_chw, layers)
    rng = np.random.default_rng(seed)
    out = {}
    first = True
    for i, ly in enumerate(layers):
        h, w, c = sizes[i]
        if ly["type"] == CONV:
            cg = c // ly["grp"]
            fan_in = ly["knl"] * ly["knl"] * cg
            shape = (ly["cnt"], cg, ly["knl"], ly["knl"])
            ct = ly["cnt"]
        elif ly["type"] == FCNT:
            fan_in = h * w * c
            shape = (ly["nod"], fan_in)
            ct = ly["nod"]
        else:
            continue
        scale = np.float32(np.sqrt(2.0 / fan_in) * (1.0 / 64.0 if first else 1.0))
        first = False
        out[i] = dict(bias=(rng.standard_normal(ct) * 0.1).astype(np.float32),
                      weights=(rng.standard_normal(shape) * scale).astype(np.float32))
    return out


def write_dense_param_dir(dir_path: str, prefix: str, params) -> None:
    os.makedirs(dir_path, exist_ok=True)
    for i, p in params.items():
        fileio.write_bin(fileio.param_path(dir_path, prefix, "biasVec", i + 1, "bin"), p["bias"])
        kind = "convKnl" if p["weights"].ndim == 4 else "fcntWei"
        fileio.write_bin(fileio.param_path(dir_path, prefix, kind, i + 1, "bin"), p["weights"])


def write_param_dir(dir_path: str, prefix: str, params) -> None:
    os.makedirs(dir_path, exist_ok=True)
    for i, p in params.items():
        fileio.write_bin(fileio.param_path(dir_path, prefix, "biasVec", i + 1, "bin"), p["bias"])
        fileio.write_bin(fileio.param_path(dir_path, prefix, "ctrdLst", i + 1, "bin"), p["ctrd"])
        fileio.write_cbn(fileio.param_path(dir_path, prefix, "asmtLst", i + 1, "cbn"), p["asmt"], p["bits"])


def load_param_dir(dir_path: str, prefix: str, layers):
    """Read a parameter directory (shipped or synthetic).  Missing files raise FileNotFoundError."""
    out = {}
    for i, ly in enumerate(layers):
        if ly["type"] not in (CONV, FCNT):
            continue
        bias = fileio.read_bin(fileio.param_path(dir_path, prefix, "biasVec", i + 1, "bin"), np.float32)
        ctrd = fileio.read_bin(fileio.param_path(dir_path, prefix, "ctrdLst", i + 1, "bin"), np.float32)
        asmt, bits = fileio.read_cbn(fileio.param_path(dir_path, prefix, "asmtLst", i + 1, "cbn"))
        out[i] = dict(bias=bias.reshape(-1), ctrd=ctrd, asmt=asmt, bits=bits)
    return out


def make_images(n: int, in_chw, seed=1234, mean: Optional[np.ndarray] = None) -> np.ndarray:
    """Synthetic input batch [n, C, H, W] fp32: uniform 8-bit pixels minus a mean image
    (the shipped 3x256x256 mean centre-cropped when given, else the ImageNet BGR channel means) —
    the value range the reference's BmpImgIO produces (src/BmpImgIO.cc:96-98,203-224)."""
    c, h, w = in_chw
    rng = np.random.default_rng(seed)
    px = rng.integers(0, 256, size=(n, c, h, w), dtype=np.uint8).astype(np.float32)
    if mean is not None:
        oh, ow = (mean.shape[1] - h) // 2, (mean.shape[2] - w) // 2
        px -= mean[None, :, oh:oh + h, ow:ow + w]
    else:
        ch_mean = np.array([104.0, 117.0, 123.0], dtype=np.float32)[:c]
        px -= ch_mean[None, :, None, None]
    return px


# ---- the shipped AlexNet parameter set + the one file the mount lacks (fc6 assignments) ----
ALEXNET_FC6 = 15                     # layer index of fc6 (file number 16)
FC6_FIXTURE2_NAME = "AlexNet/fixtures/bvlc_alexnet_aCaF.asmtLst.16.fx2.cbn"


def fc6_fixture(ctrd: np.ndarray, fixture: int) -> np.ndarray:
    """Synthetic fc6 assignment matrix [4096, M] (0-based, < K) for the shipped fc6 code book ``ctrd`` [M, K, Cs].

    fixture 1  SURVEY.md §8c recipe: default_rng(0).integers(0, K).  With it the real network's tail degenerates (fc7
               all-negative, fc8 = bias, every image the same top-5), so it cannot test fc7 / fc8 / top-5.
    fixture 2  default_rng(2) picks, per (channel, sub-space), one of the EIGHT code words of smallest norm: fc6
               activations stay in the range the trained fc7 expects (28 % positive, max ~12), fc7 keeps ~14 % of its
               units alive, fc8 and the top-5 differ from image to image (7 distinct top-1 over the 10 shipped BMPs).
    """
    m, k, _ = ctrd.shape
    if fixture == 1:
        return np.random.default_rng(0).integers(0, k, size=(4096, m), dtype=np.uint8)
    if fixture != 2:
        raise ValueError("fc6 fixture %r" % (fixture,))
    order = np.argsort(np.sqrt((ctrd.astype(np.float64) ** 2).sum(-1)), axis=1, kind="stable")   # [M, K], smallest norm first
    pick = np.random.default_rng(2).integers(0, 8, size=(4096, m))
    return order[np.arange(m)[None, :], pick].astype(np.uint8)


def load_alexnet_shipped(data_root: str, layers, fixture: int = 1):
    """The shipped AlexNet parameters from a staged data root (oracle/_ref/data: AlexNet/Bin.Files/*), fc6 assignments =
    the staged fixture 1 file or (fixture 2) the staged / regenerated non-degenerate one."""
    params = load_param_dir(os.path.join(data_root, "AlexNet/Bin.Files"), "bvlc_alexnet_aCaF", layers)
    if fixture == 2:
        path = os.path.join(data_root, FC6_FIXTURE2_NAME)
        asmt = fileio.read_cbn(path)[0] if os.path.exists(path) else fc6_fixture(params[ALEXNET_FC6]["ctrd"], 2)
        params[ALEXNET_FC6] = dict(params[ALEXNET_FC6], asmt=asmt)
    return params


def shipped_mean_image(data_root: str) -> np.ndarray:
    """AlexNet/imagenet_mean.single.bin [3, 256, 256] (BGR) of a staged data root."""
    return fileio.read_bin(os.path.join(data_root, "AlexNet/imagenet_mean.single.bin"), np.float32)
