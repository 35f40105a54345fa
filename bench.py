#!/usr/bin/env python3
"""bench.py — images/second of the Quantized-CNN approximate forward pass on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One JSON line on stdout (rank 0).

A step = one pass of the hot path (pack -> 23 AlexNet layers, LUT build + indexed accumulation for
conv/FC, glue layers, top-5) over one batch of --batch synthetic images PER GPU, inputs already
resident in HBM.  Images are independent: ranks share nothing on the data path; the only collective
is the one-time RCCL broadcast of rank 0's packed parameter arena (codebooks, assignments, biases),
outside the timed region.  Scaling is therefore "weak" (per-GPU work fixed).

Extra objects on the JSON line:
  roofline      dominant kernel (largest mean HIP-event time over the timed steps), algorithmic HBM
                bytes per launch / its duration against 8 TB/s; `traffic` = that kernel's HBM bytes from
                the committed rocprofv3 --pmc passes (profiles/); plus the LDS look-up rate against the
                ds_read_b64 peak, the resource that actually bounds the kernel.
  cpu_baseline  the reference itself (oracle/_ref/libqcnn_ref.so, kind "reference") — or the C port
                when that was never built — timed single-threaded on this host on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def pkg(name=""):
    return importlib.import_module("quantized-cnn_amd" + ("." + name if name else ""))


HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LDS_B64_LOOKUPS_PER_S = 256 * 64 * 2.4e9    # 256 CU x 256 B/clk (ds_read_b64, an image pair per lane) / 4 B x 2.4 GHz
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r1_v6", "traffic.json")   # PMC HBM bytes of the dominant kernel


def pmc_traffic(layer, launches_per_forward):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE + WRITE_SIZE, separate passes, mean per dispatch), or None when no profile covers it."""
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        if int(t["layer"]) != int(layer) or int(t.get("launches_per_forward", 1)) != int(launches_per_forward):
            return None
        return int(t["bytes"])
    except (OSError, ValueError, KeyError):
        return None


def algorithmic_bytes(sizes, layers, params, l, batch, fused):
    """HBM bytes one launch of layer l must move: read fm[l], write fm[l+1], read its parameters once
    (SURVEY.md §8d 'A_layer').  The look-up table never leaves the CU."""
    topo = pkg("topology")
    e_in = sizes[l][0] * sizes[l][1] * sizes[l][2]
    e_out = sizes[l + 1][0] * sizes[l + 1][1] * sizes[l + 1][2]
    b = 4.0 * batch * (e_in + e_out)
    if l in params:
        p = params[l]
        b += p["bias"].nbytes + p["ctrd"].nbytes + p["asmt"].nbytes
    return b


def lookups_per_image(sizes, layers, params, l):
    """Border-clipped table look-ups of layer l per image (the reference's trip count)."""
    topo = pkg("topology")
    ly = layers[l]
    if ly["type"] == topo.FCNT:
        return params[l]["ctrd"].shape[0] * ly["nod"]
    if ly["type"] != topo.CONV:
        return 0
    h, w, _ = sizes[l]
    ho, wo, ct = sizes[l + 1]
    k, s, p = ly["knl"], ly["stride"], ly["pad"]
    taps_h = sum(min(k - 1, h - 1 - (o * s - p)) - max(0, -(o * s - p)) + 1 for o in range(ho))
    taps_w = sum(min(k - 1, w - 1 - (o * s - p)) - max(0, -(o * s - p)) + 1 for o in range(wo))
    return taps_h * taps_w * params[l]["ctrd"].shape[0] * ct


def cpu_baseline(in_chw, layers, params, imgs_host, sample):
    """Time the CPU path on this host: the compiled reference if present, else the C port."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    synth = pkg("synth")
    imgs = imgs_host[:sample]
    if po.have_ref():
        with tempfile.TemporaryDirectory() as d:
            synth.write_param_dir(d, "bench", params)
            ref = po.RefLib()
            ref.load_custom(d, "bench", in_chw, layers)
        ref.time_forward(imgs[:2])                      # page in
        wall, cpu = ref.time_forward(imgs)
        return dict(value=sample / cpu, unit="images/s", cores=1, kind="reference",
                    sample="%d images, batch 1 (the reference's own regime), single thread; %.2f s CPU time by the "
                           "reference's swAllLayers stop-watch (clock()), %.2f s wall; g++ -O2 Makefile.native flags "
                           "(ATLAS/OpenVML not installed)" % (sample, cpu, wall),
                    host_cores=os.cpu_count(), ms_per_image=1000.0 * cpu / sample)
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    orc.forward(imgs[:1])
    t0 = time.perf_counter()
    for i in range(sample):
        orc.forward(imgs[i:i + 1])
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="images/s", cores=1, kind="port",
                sample="%d images, batch 1, single thread, oracle/qcnn_oracle.c -O2; %.2f s wall" % (sample, dt),
                host_cores=os.cpu_count(), ms_per_image=1000.0 * dt / sample)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1000, help="images per GPU per step")
    ap.add_argument("--model", default="AlexNet")
    ap.add_argument("--lut", default="mfma", choices=["mfma", "exact"])
    ap.add_argument("--cpu-sample", type=int, default=100, help="images for the CPU baseline (0 = skip)")
    ap.add_argument("--h2d-steps", type=int, default=2, help="extra steps timed including pinned-host H2D (0 = skip)")
    ap.add_argument("--streams", type=int, default=1,
                    help="sub-batches of whole panels run concurrently on separate HIP streams (QCNN_OPT_STREAMS; the "
                         "library default is 2).  1 keeps one launch per layer, so that the per-kernel HIP-event "
                         "durations behind `roofline` are those of kernels that own the whole GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    topo, synth, capi = pkg("topology"), pkg("synth"), pkg("capi")
    in_chw, layers, _, _ = topo.MODELS[args.model]
    sizes = topo.fmap_sizes(in_chw, layers)
    params = synth.make_params(in_chw, layers, seed=0)      # every rank knows the SHAPES; rank 0 owns the VALUES
    B = args.batch

    # one explicit stream for everything (torch ops, H2D copies and the engine's kernels): the default stream's
    # handle is NULL, which the C-ABI reads as "create your own stream" — the copies would then not be ordered
    # with the forward passes
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = pkg("engine").QcnnEngine(local, stream.cuda_stream)
    eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA if args.lut == "mfma" else capi.LUT_EXACT)
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.set_option(capi.OPT_PROFILE, 1)
    eng.set_option(capi.OPT_STREAMS, args.streams)
    shapes = {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()}
    eng.configure(in_chw, layers, shapes)
    arena = torch.zeros(eng.arena_bytes(), dtype=torch.uint8, device=dev)
    eng.commit(B, arena.data_ptr())
    if rank == 0:
        eng.upload(params)
    if world > 1:
        dist.broadcast(arena, src=0)                        # RCCL over xGMI: codebooks + assignments + biases
        torch.cuda.synchronize(dev)
    if rank != 0:
        eng.mark_loaded()

    # synthetic device-resident input: 8-bit pixels minus the BGR channel means (range of BmpImgIO's output)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    imgs = torch.randint(0, 256, (B,) + tuple(in_chw), generator=g, device=dev, dtype=torch.int32).to(torch.float32)
    imgs -= torch.tensor([104.0, 117.0, 123.0], device=dev)[: in_chw[0]].view(1, -1, 1, 1)
    classes = sizes[-1][0] * sizes[-1][1] * sizes[-1][2]
    prob = torch.empty((B, classes), dtype=torch.float32, device=dev)
    top5 = torch.empty((B, 5), dtype=torch.int16, device=dev)

    def step():
        eng.forward_dev(imgs.data_ptr(), B, prob.data_ptr(), top5.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    eng.reset_layer_ms()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    layer_ms, recorded = eng.layer_ms()
    ok = bool(torch.isfinite(prob).all().item())

    h2d = h2d_u8 = two_streams = None
    if args.h2d_steps > 0 and rank == 0 and args.streams == 1:
        # the library's default execution mode: two sub-batches on two streams (glue kernels of one overlap the
        # conv tails of the other); reported next to `value`, which stays the one-launch-per-layer measurement
        eng.set_option(capi.OPT_STREAMS, 2)
        eng.set_option(capi.OPT_PROFILE, 0)
        step()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        for _ in range(max(args.h2d_steps, 3)):
            step()
        torch.cuda.synchronize(dev)
        two_streams = B * max(args.h2d_steps, 3) / (time.perf_counter() - t3)
        eng.set_option(capi.OPT_STREAMS, 1)
    if args.h2d_steps > 0 and rank == 0:
        pinned = torch.empty(imgs.shape, dtype=torch.float32, pin_memory=True)
        pinned.copy_(imgs)
        staging = torch.empty_like(imgs)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.h2d_steps):
            staging.copy_(pinned, non_blocking=True)
            eng.forward_dev(staging.data_ptr(), B, prob.data_ptr(), top5.data_ptr())
        torch.cuda.synchronize(dev)
        h2d = B * args.h2d_steps / (time.perf_counter() - t1)
        # same with the device-side input pipeline: 8-bit 256x256 source images + mean image, crop on the device
        hs, ws = max(in_chw[1], 256), max(in_chw[2], 256)
        px = torch.randint(0, 256, (B, in_chw[0], hs, ws), dtype=torch.uint8)
        pinned_u8 = torch.empty(px.shape, dtype=torch.uint8, pin_memory=True)
        pinned_u8.copy_(px)
        staging_u8 = torch.empty(px.shape, dtype=torch.uint8, device=dev)
        mean_img = torch.full((in_chw[0], hs, ws), 110.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(args.h2d_steps):
            staging_u8.copy_(pinned_u8, non_blocking=True)
            eng.forward_u8_dev(staging_u8.data_ptr(), hs, ws, mean_img.data_ptr(), B, prob.data_ptr(), top5.data_ptr())
        torch.cuda.synchronize(dev)
        h2d_u8 = B * args.h2d_steps / (time.perf_counter() - t2)

    if rank == 0:
        ms_step = 1000.0 * dt / args.steps
        value = world * B * args.steps / dt
        dom = int(np.argmax(layer_ms))
        dom_ms = float(layer_ms[dom])
        # a layer is launched once per sub-batch (QCNN_OPT_STREAMS): layer_ms is the mean duration of ONE launch,
        # so the algorithmic bytes / look-ups are those of one launch (its share of the panels)
        panels = (B + 127) // 128
        ns = max(1, min(args.streams, panels))
        launch_images = B / ns
        abytes = algorithmic_bytes(sizes, layers, params, dom, launch_images, True)
        achieved = abytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        lk = lookups_per_image(sizes, layers, params, dom) * launch_images
        total_lk = sum(lookups_per_image(sizes, layers, params, l) for l in range(len(layers)))
        conv_idx = [i for i, l in enumerate(layers) if l["type"] == topo.CONV]
        name = "%s%d" % (topo.TYPE_NAMES[layers[dom]["type"]], (conv_idx.index(dom) + 1) if dom in conv_idx else dom)
        roof = dict(bound="hbm", kernel="k_%s_aprx (layer %d, %s)" % ("conv" if dom in conv_idx else "fc", dom, name),
                    achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5),
                    traffic=(pmc_traffic(dom, ns) if (B == 1000 and args.model == "AlexNet") else None),
                    ms_per_launch=round(dom_ms, 4), launches_timed=recorded * ns, launches_per_step=ns,
                    images_per_launch=launch_images,
                    algorithmic_bytes_per_launch=int(abytes),
                    lds_lookups_per_s=round(lk / (dom_ms * 1e-3), 0) if dom_ms > 0 else 0,
                    lds_lookup_peak_b64=LDS_B64_LOOKUPS_PER_S,
                    lds_frac=round(lk / (dom_ms * 1e-3) / LDS_B64_LOOKUPS_PER_S, 4) if dom_ms > 0 else 0,
                    layer_ms={"%02d_%s" % (i, topo.TYPE_NAMES[layers[i]["type"]]): round(float(m), 4)
                              for i, m in enumerate(layer_ms) if m > 0})
        out = {
            "metric": "images/sec AlexNet quantized forward" if args.model == "AlexNet" else "images/sec %s quantized forward" % args.model,
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s Q-CNN approximate forward (fp32 LUT + uint8 indices), %d synthetic %dx%d images per GPU per step, "
                                   "device-resident" % (args.model, B, in_chw[1], in_chw[2]),
                       "images_per_gpu": B, "global_batch": B * world, "lut_builder": args.lut,
                       "parameters": "seeded synthetic, shipped AlexNet quantisation shapes",
                       "streams_per_gpu": ns,
                       "parallelism": "images sharded over %d GPU(s), parameters replicated by one RCCL broadcast" % world},
            "outputs_finite": ok,
            "lookups_per_image": int(total_lk),
            "lookups_per_s": round(total_lk * value, 0),
            "roofline": roof,
        }
        if h2d is not None:
            out["value_incl_pinned_h2d"] = round(h2d, 2)
            out["value_incl_pinned_h2d_u8"] = round(h2d_u8, 2)
        if two_streams is not None:
            out["value_two_streams"] = round(two_streams, 2)
        if args.cpu_sample > 0 and world == 1:
            imgs_host = imgs[: args.cpu_sample].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(in_chw, layers, params, imgs_host, args.cpu_sample)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
