"""CPU tier: the pieces of bench.py that do not need a GPU — the committed profile summary it quotes (`roofline.traffic`,
`roofline.rocprof_avg_ms_per_launch`) is readable and belongs to the device sources in the tree, the per-layer work model covers every
conv / FC layer of the two benchmarked topologies, the algorithmic figures are SURVEY.md §8d's."""
import importlib.util
import json
import os

import numpy as np

from conftest import ROOT, pkg

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
topo, synth, perf = pkg("topology"), pkg("synth"), pkg("perfmodel")


def test_committed_traffic_summary_matches_the_device_sources():
    """profiles/<newest>/traffic.json carries the hash of quantized-cnn_amd/csrc it was measured on; bench.py quotes its HBM bytes and
    rocprofv3 kernel averages only for that build.  A change to a device source without a new profile run shows up here."""
    prof = os.path.join(ROOT, "profiles")
    newest = sorted(d for d in os.listdir(prof) if os.path.exists(os.path.join(prof, d, "traffic.json")))[-1]
    t = json.load(open(os.path.join(prof, newest, "traffic.json")))
    assert t["kernel_hash"] == bench.kernel_hash(), "%s/traffic.json was taken from another build: re-run scripts/gpu_prof.sh" % newest
    bytes_, src, table = bench.pmc_traffic(int(t["layer"]), 1)
    assert bytes_ == int(t["bytes"]) and src.endswith("traffic.json")
    dom = table[int(t["layer"])]
    assert dom["rocprof_avg_ms"] > 0 and dom["kernel"].startswith("k_conv")
    # the profiled average and the HIP-event time of the same run agree to a few per cent ("must agree")
    assert abs(dom["rocprof_avg_ms"] / dom["hip_event_ms_under_rocprof"] - 1.0) < 0.08


def test_work_model_covers_both_benchmarked_topologies():
    for model, want_lookups, want_mac in (("AlexNet", 116276576, 51387776), ("VGG16", 1907710208, 1163476992)):
        in_chw, layers, _, _ = topo.MODELS[model]
        sizes = topo.fmap_sizes(in_chw, layers)
        spec_ = synth.quant_spec(in_chw, layers)
        lookups, mac = 0, 0
        for i, l in enumerate(layers):
            if l["type"] == topo.CONV:
                w = perf.conv_work(sizes[i], sizes[i + 1], l, spec_[i]["M"], spec_[i]["K"], spec_[i]["Cs"])
                lookups += w["lookups"]
                mac += w["alg_flop"] // 2 // 128
            elif l["type"] == topo.FCNT:
                lookups += spec_[i]["M"] * l["nod"]
                d = sizes[i][0] * sizes[i][1] * sizes[i][2]
                mac += spec_[i]["K"] * d
        assert lookups == want_lookups, (model, lookups)              # gathers per image, border-clipped
        assert mac == want_mac, (model, mac)                          # ... and LUT-build multiply-adds per image


def test_layer_report_for_every_kernel_family():
    """perfmodel.layer_report (the `roofline.layers` entries) for the tile, 16-wave sliding, eight-wave tile and eight-wave sliding
    forms of AlexNet's and VGG-16's conv layers and for the FC kernels: finite, positive figures."""
    for model in ("AlexNet", "VGG16"):
        in_chw, layers, _, _ = topo.MODELS[model]
        sizes = topo.fmap_sizes(in_chw, layers)
        spec_ = synth.quant_spec(in_chw, layers)
        params = {i: dict(ctrd=np.zeros((s["M"], s["K"], s["Cs"]), np.float32)) for i, s in spec_.items()}
        for i, l in enumerate(layers):
            if l["type"] not in (topo.CONV, topo.FCNT):
                continue
            variants = [dict()] if l["type"] == topo.FCNT else [dict(), dict(sym=8)]
            if l["type"] == topo.CONV and l["knl"] == 3 and sizes[i][2] // l["grp"] >= 64 and l["cnt"] // l["grp"] >= 128:
                variants += [dict(seg_beg=[0, sizes[i + 1][0]], sym=8), dict(seg_beg=[0, sizes[i + 1][0] // 2, sizes[i + 1][0]], sym=8)]
            if l["type"] == topo.CONV and l["cnt"] // l["grp"] in (128, 192, 256, 384, 512) and sizes[i][2] // l["grp"] >= 8:
                variants += [dict(sym="h8")]                   # half-panel eight-wave kernel (round 6), tile form ...
                if l["knl"] == 3:                              # ... and sliding form
                    variants += [dict(seg_beg=[0, sizes[i + 1][0] // 2, sizes[i + 1][0]], sym="h8")]
            for kw in variants:
                r = perf.layer_report(sizes, layers, params, i, 1000.0, 1.5, **kw)
                if kw.get("sym") == "h8" and not kw.get("seg_beg"):
                    # twice the tile: fewer table builds per source pixel than the full-panel eight-wave tile
                    r8 = perf.layer_report(sizes, layers, params, i, 1000.0, 1.5, sym=8)
                    assert r["rebuild_factor"] < r8["rebuild_factor"], (model, i, r["rebuild_factor"], r8["rebuild_factor"])
                for key in ("stages_per_panel", "rebuild_factor", "lookups_per_stage", "stage_cycles", "lds_frac"):
                    assert np.isfinite(r[key]) and r[key] > 0, (model, i, kw, key, r)
