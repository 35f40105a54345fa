#!/usr/bin/env python3
"""bench.py — images/second of the Quantized-CNN approximate forward pass on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches it under
torch.distributed.run (one rank per GPU); started WITHOUT that environment, `--gpus N` re-executes itself under
torch.distributed.run and fails loudly when fewer than N GPUs are visible.  One JSON line on stdout (rank 0).

A step = one pass of the hot path (23 AlexNet layers: LUT build + indexed accumulation for conv/FC, glue layers,
top-5) over one batch of synthetic images already resident in HBM.
  N = 1   the batch is BASELINE.json configs[1]: 1000 images on one GPU.
  N > 1   configs[2]: ONE 1000-image batch sharded over the N GPUs in contiguous blocks (dist.shard_bounds):
          "scaling": "strong".  Images are independent, so there is no data-path collective; the only exchange is
          the one-time RCCL broadcast of rank 0's packed parameter arena, outside the timed region (its time is
          reported as `param_broadcast_ms`).  `value_weak` additionally reports 1000 images PER GPU.  Before anything is timed
          every rank hashes its arena on the device and the ranks compare the pairs over the communicator
          (`param_broadcast_verified`); `rccl_ranks` is counted by a real all_reduce; a watchdog (`--init-timeout`) turns a hung
          rendezvous into one JSON error line.  `--dry-run-shared-gpu 1` runs this whole path with every rank on GPU 0 and the
          collectives over gloo (labelled `dry_run`, not a measurement): how a one-GPU box exercises it.

`value` runs the library defaults on ONE stream (so that a layer's HIP-event time is that layer's alone): conv1 and fc8 through the
code words their assignments name (`config.decoded_layers`).  Right behind it: `alg_north_star_value` / `_ms_per_step` /
`_roofline_frac` — the north star's algorithm, look-up tables + uint8-indexed accumulation, on EVERY conv / FC layer (= value_tables_only).
`value_two_streams` is the library's default execution mode (two sub-batches on two streams).

Extra objects on the JSON line:
  roofline      dominant kernel (largest mean HIP-event time per launch over the timed steps), algorithmic HBM bytes
                per launch / that time against 8 TB/s; `traffic` = its HBM bytes from the committed rocprofv3 --pmc
                passes when they were taken from THIS kernel source (hash check), else null; per conv/FC layer the
                figures the kernels are really bounded by (stages built, rebuild factor, cycles per stage, look-ups
                against the LDS read peak, matrix-pipe utilisation).
  parity        images of the timed batch — half from the first panel, half from the end of the last, ragged one — run
                through the CPU reference (or the C port) and compared with what the GPU produced for them:
                probabilities, pool5 feature map, top-5.  > 1e-4 aborts.
  cpu_baseline  the reference itself (oracle/_ref/libqcnn_ref.so, kind "reference") — or the C port when that was
                never built — timed single-threaded on this host on a bounded sample (N = 1 only).
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def pkg(name=""):
    return importlib.import_module("quantized-cnn_amd" + ("." + name if name else ""))


HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
TOL = 1e-4                                  # north_star tolerance, relative to the map's largest magnitude
KERNEL_DIR = os.path.join(ROOT, "quantized-cnn_amd", "csrc")


def kernel_hash():
    """Hash of every device source (kernels, glue, few-image kernels, engine, group, their header): a committed PMC
    traffic figure is only quoted for the build it was taken from."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(KERNEL_DIR)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(KERNEL_DIR, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(layer, launches_per_forward):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 --pmc summary
    (profiles/*/traffic.json: FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE, separate passes,
    mean per dispatch) — only if it was taken from the kernel source that is being benchmarked.  Returns (bytes or None,
    source / reason, {layer: entry} of every kernel the summary covers)."""
    prof = os.path.join(ROOT, "profiles")
    best = None
    for d in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        p = os.path.join(prof, d, "traffic.json")
        if os.path.exists(p):
            best = p
    if best is None:
        return None, "no profiles/*/traffic.json", {}
    try:
        with open(best) as f:
            t = json.load(f)
        rel = os.path.relpath(best, ROOT)
        if t.get("kernel_hash") != kernel_hash():
            return None, "%s was taken from another build of quantized-cnn_amd/csrc (hash mismatch)" % rel, {}
        if int(t.get("launches_per_forward", 1)) != int(launches_per_forward):
            return None, "%s was taken with %s launches per forward" % (rel, t.get("launches_per_forward")), {}
        table = {int(k): v for k, v in t.get("kernels", {}).items()}
        if "layer" in t and int(t["layer"]) not in table:
            table[int(t["layer"])] = dict(kernel=t.get("kernel"), bytes=int(t["bytes"]))
        if int(layer) not in table:
            return None, "%s covers layers %s" % (rel, sorted(table)), table
        return int(table[int(layer)]["bytes"]), rel, table
    except (OSError, ValueError, KeyError) as e:
        return None, "unreadable: %s" % e, {}


def algorithmic_bytes(sizes, layers, params, l, batch):
    """HBM bytes one launch of layer l must move: read fm[l], write fm[l+1], read its parameters once
    (SURVEY.md §8d 'A_layer').  The look-up table never leaves the CU."""
    e_in = sizes[l][0] * sizes[l][1] * sizes[l][2]
    e_out = sizes[l + 1][0] * sizes[l + 1][1] * sizes[l + 1][2]
    b = 4.0 * batch * (e_in + e_out)
    if l in params:
        p = params[l]
        b += p["bias"].nbytes + p["ctrd"].nbytes + p["asmt"].nbytes
    return b


def cpu_side(in_chw, layers, params):
    """The CPU checker: the compiled reference when present, else the C port.  Returns (kind, forward(img)->
    (prob, pool-map getter), timer)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    synth = pkg("synth")
    if po.have_ref():
        with tempfile.TemporaryDirectory() as d:
            synth.write_param_dir(d, "bench", params)
            ref = po.RefLib()
            ref.load_custom(d, "bench", in_chw, layers)
        return "reference", ref, po
    orc = po.COracle(in_chw, layers)
    orc.set_params(params)
    return "port", orc, po


def parity_check(kind, cpu, layers, imgs_host, gpu_prob, gpu_top5, gpu_fms):
    """Compare the GPU's outputs for imgs_host with the CPU checker's: soft-max outputs, top-5 and the feature maps
    gpu_fms = {index: [n, ...]}.  Error = max |a-b| / max |b| per map and image."""
    worst_prob = 0.0
    worst_fm = {l: 0.0 for l in gpu_fms}
    agree = 0
    distinct = set()
    n = imgs_host.shape[0]
    for i in range(n):
        if kind == "reference":
            prob = cpu.forward(imgs_host[i:i + 1])
            top5 = cpu.top5()
        else:
            cpu.forward(imgs_host[i:i + 1])
            prob = cpu.fm(len(layers)).reshape(-1)
            top5 = cpu.top5(prob)
        for l in gpu_fms:
            fm = cpu.fm(l)[0]
            worst_fm[l] = max(worst_fm[l], float(np.abs(gpu_fms[l][i] - fm).max() / max(np.abs(fm).max(), 1e-30)))
        worst_prob = max(worst_prob, float(np.abs(gpu_prob[i] - prob).max() / max(np.abs(prob).max(), 1e-30)))
        agree += int(np.array_equal(np.asarray(gpu_top5[i], np.uint16), np.asarray(top5, np.uint16)))
        distinct.add(tuple(int(x) for x in top5))
    return dict(images=n, checker=kind, top5_agree=agree, distinct_top5=len(distinct), max_rel_err_prob=worst_prob,
                max_rel_err_fm=max(worst_fm.values()) if worst_fm else 0.0,
                fm_checked={str(l): v for l, v in worst_fm.items()}, tolerance=TOL,
                ok=bool(agree == n and worst_prob <= TOL and all(v <= TOL for v in worst_fm.values())))


def cpu_workers(so_path, pdir, n_proc, n_img, lead_s):
    """n_proc processes of oracle/ref_worker.py (the reference is single-threaded and not re-entrant: one process per
    core), all starting their timed forwards at the same wall-clock instant.  Returns (images/s over the window from the
    first start to the last finish, per-process records) or None."""
    worker = os.path.join(ROOT, "oracle", "ref_worker.py")
    start = time.time() + lead_s
    procs = [subprocess.Popen([sys.executable, worker, so_path, pdir, "bench", str(n_img), repr(start)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(n_proc)]
    recs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=lead_s + 120)
            recs.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            pr.kill()
    if len(recs) != n_proc:
        return None
    t0, t1 = min(r["t0"] for r in recs), max(r["t1"] for r in recs)
    return sum(r["n"] for r in recs) / (t1 - t0), recs


def cpu_baseline(kind, cpu, imgs_host, params=None):
    sample = imgs_host.shape[0]
    if kind == "reference":
        cpu.time_forward(imgs_host[:2])                      # page in
        wall, cpu_s = cpu.time_forward(imgs_host)
        cb = dict(value=sample / cpu_s, unit="images/s", cores=1, kind="reference",
                  sample="%d images, batch 1 (the reference's own regime), single thread; %.2f s CPU time by the "
                         "reference's swAllLayers stop-watch (clock()), %.2f s wall; g++ -O2 Makefile.native flags "
                         "(ATLAS/OpenVML not installed)" % (sample, cpu_s, wall),
                  host_cores=os.cpu_count(), ms_per_image=1000.0 * cpu_s / sample)
        # stronger comparators beside the stated baseline: the same sources at -O3 -march=znver3, and one
        # process of the -O2 build per core (the reference is single-threaded by construction)
        import pyoracle as po
        if params is not None:
            try:
                with tempfile.TemporaryDirectory() as d:
                    pkg("synth").write_param_dir(d, "bench", params)
                    if os.path.exists(po.REF_O3_SO):
                        r = cpu_workers(po.REF_O3_SO, d, 1, 40, 6.0)
                        if r:
                            cb["o3"] = dict(value=round(r[1][0]["n"] / r[1][0]["cpu"], 2), unit="images/s", cores=1,
                                            build="g++ -O3 -march=znver3 (oracle/Makefile ref_o3; the host is Zen 5, gcc 11.4 "
                                                  "knows no newer AMD target)", sample="40 images, one process")
                    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                    per = 12
                    r = cpu_workers(po.REF_SO, d, ncpu, per, 8.0 + 0.05 * ncpu)
                    if r:
                        cb["all_cores"] = dict(value=round(r[0], 1), unit="images/s", cores=ncpu, processes=ncpu,
                                               sample="%d processes (one per logical core) x %d images of the -O2 reference, common "
                                                      "start; images / (last finish - first start)" % (ncpu, per),
                                               mean_ms_per_image_per_process=round(1000.0 * sum(x["wall"] for x in r[1]) / (ncpu * per), 2))
            except Exception as e:                           # comparators only: never fail the bench line
                cb["extras_error"] = repr(e)[:200]
        return cb
    cpu.forward(imgs_host[:1])
    t0 = time.perf_counter()
    for i in range(sample):
        cpu.forward(imgs_host[i:i + 1])
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="images/s", cores=1, kind="port",
                sample="%d images, batch 1, single thread, oracle/qcnn_oracle.c -O2; %.2f s wall" % (sample, dt),
                host_cores=os.cpu_count(), ms_per_image=1000.0 * dt / sample)


def via_reference_main(params, imgs_host, batches=16):
    """images/s of the reference's UNMODIFIED src/Main.cc + src/UnitTest.cc (build/bin/QuanCNN_hip, linked against the
    host mirror) on a staged data root holding this run's parameters and images: QCNN_BATCH = len(imgs) images per
    forward pass, `batches` passes over the same window (the reference's window rule for a one-batch dataset,
    src/CaffeEva.cc:170-177).  The rate is images / the host wall clock around ExecForwardPass(void) (printed by
    DispElpsTime as swDebugTimePri: uploads from the pinned dataset + kernels + top-5 read-back); the program's own
    `elapsed time` (dataset and parameter files, device set-up, warm-up included) is reported beside it."""
    exe = os.path.join(ROOT, "build", "bin", "QuanCNN_hip")
    if not os.path.exists(exe):
        return None
    import re
    synth, fileio = pkg("synth"), pkg("fileio")
    n = imgs_host.shape[0]
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "AlexNet", "Bin.Files"))
        os.makedirs(os.path.join(root, "ILSVRC12.227x227.IMG"))
        synth.write_param_dir(os.path.join(root, "AlexNet", "Bin.Files"), "bvlc_alexnet_aCaF", params)
        fileio.write_bin(os.path.join(root, "ILSVRC12.227x227.IMG", "dataMatTst.single.bin"), imgs_host)
        fileio.write_bin(os.path.join(root, "ILSVRC12.227x227.IMG", "lablVecTst.uint16.bin"),
                         np.zeros((1, 1, 1, n), np.uint16))
        env = dict(os.environ, QCNN_BATCH=str(n), QCNN_BATCHES=str(batches), QCNN_MAX_INFLIGHT=str(max(n, 1024)),
                   QCNN_DEVICE="0")
        try:
            r = subprocess.run([exe], cwd=root, capture_output=True, text=True, timeout=900, env=env)
        except subprocess.TimeoutExpired:
            return dict(error="timeout")
    out = r.stdout
    wall = re.search(r"swDebugTimePri: ([0-9.]+)", out)
    dev = re.search(r"swAllLayers: ([0-9.]+)", out)
    tot = re.search(r"elapsed time: ([0-9.]+)", out)
    if r.returncode != 0 or not wall or float(wall.group(1)) <= 0 or out.count("processing the ") != batches:
        return dict(error="rc=%d" % r.returncode, tail=out[-400:])
    return dict(value=round(n * batches / float(wall.group(1)), 2), unit="images/s", images=n * batches,
                exec_forward_pass_wall_s=float(wall.group(1)), device_layers_s=float(dev.group(1)) if dev else None,
                program_elapsed_s=float(tot.group(1)) if tot else None,
                how="QCNN_BATCH=%d QCNN_BATCHES=%d build/bin/QuanCNN_hip (byte-identical reference Main.cc/UnitTest.cc); "
                    "rate = images / swDebugTimePri" % (n, batches))


def self_spawn(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.dry_run_shared_gpu and have >= 1:
        have = args.gpus                 # dry run of the N > 1 code path: every rank on GPU 0, collectives over gloo (never a result)
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible — refusing to report a %d-GPU number"
                         % (args.gpus, have, args.gpus))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def timed(torch, dev, fn, steps):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def group_capi_child(args):
    """Child process of the N = 1 bench: one batch through qcnn_group_forward over every visible device (see the call site)."""
    import torch
    topo, synth, capi = pkg("topology"), pkg("synth"), pkg("capi")
    in_chw, layers, _, _ = topo.MODELS[args.model]
    data_root = os.path.join(ROOT, "oracle", "_ref", "data")
    have_shipped = args.model == "AlexNet" and os.path.exists(os.path.join(data_root, "AlexNet/Bin.Files/bvlc_alexnet_aCaF.ctrdLst.01.bin"))
    shipped = have_shipped and args.params != "synthetic"
    params = synth.load_alexnet_shipped(data_root, layers, fixture=1) if shipped else synth.make_params(in_chw, layers, seed=0)
    B = args.batch
    if shipped:
        imgs = torch.from_numpy(synth.make_images(B, in_chw, seed=1234, mean=synth.shipped_mean_image(data_root)))
    else:
        g = torch.Generator(device="cuda:0")
        g.manual_seed(1234)
        imgs = torch.randint(0, 256, (B,) + tuple(in_chw), generator=g, device="cuda:0", dtype=torch.int32).to(torch.float32)
        imgs -= torch.tensor([104.0, 117.0, 123.0], device="cuda:0")[: in_chw[0]].view(1, -1, 1, 1)
    sizes = topo.fmap_sizes(in_chw, layers)
    classes = sizes[-1][0] * sizes[-1][1] * sizes[-1][2]
    ndev = torch.cuda.device_count()
    grp = pkg("engine").QcnnDeviceGroup(list(range(ndev)))
    grp.set_option(capi.OPT_KEEP_ALL, 0)
    grp.set_option(capi.OPT_STREAMS, args.streams)
    grp.load_model(in_chw, layers, params, B)
    gb = [grp.shard_bounds(B, r) for r in range(grp.size)]
    gx = [imgs[a:b_].to("cuda:%d" % r) for r, (a, b_) in enumerate(gb)]
    gp = [torch.empty((b_ - a, classes), dtype=torch.float32, device="cuda:%d" % r) for r, (a, b_) in enumerate(gb)]
    gt = [torch.empty((b_ - a, 5), dtype=torch.int16, device="cuda:%d" % r) for r, (a, b_) in enumerate(gb)]
    for r in range(ndev):
        torch.cuda.synchronize(r)

    def gstep():
        grp.forward_dev([t.data_ptr() for t in gx], B, [t.data_ptr() for t in gp], [t.data_ptr() for t in gt])
    gstep(); gstep()
    grp.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        gstep()
    grp.sync()
    value = round(B * 5 / (time.perf_counter() - t0), 2)
    want = np.load(args.group_capi_child)
    same = all(bool(np.array_equal(gt[r].cpu().numpy(), want[a:b_])) for r, (a, b_) in enumerate(gb))
    out = dict(value=value, devices=grp.size, param_broadcast_ms=round(float(grp.broadcast_ms), 3),
               arena_checksum="%016x:%016x" % grp.arena_checksum(), top5_equal_to_single_context=same,
               note="qcnn_group_forward over every visible device of ONE process (a child of the bench, time-limited), 5 steps of one "
                    "%d-image batch, inputs resident on their ranks' devices" % B)
    grp.close()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1000, help="images of one (global) batch")
    ap.add_argument("--model", default="AlexNet")
    ap.add_argument("--lut", default="mfma", choices=["mfma", "exact", "f16"],
                    help="LUT builder: f32 MFMA (default), exact VALU (bit-identical conv/FC), f16 = opt-in fp16 table storage (configs[4] study)")
    ap.add_argument("--params", default="auto", choices=["auto", "shipped", "synthetic"],
                    help="AlexNet parameters: shipped = the reference's own files staged under oracle/_ref/data (+ the fc6 table the "
                         "mount lacks, SURVEY.md §8c fixture 1) and inputs minus the shipped mean image (SURVEY.md §8d); synthetic = "
                         "seeded, same shapes; auto = shipped where the staged files exist")
    ap.add_argument("--vgg-batch", type=int, default=1000, help="images of the VGG-16 run (BASELINE configs[3]; SURVEY.md §8d: 1000)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = one --batch sharded over the GPUs (BASELINE configs[2]); weak = --batch per GPU")
    ap.add_argument("--cpu-sample", type=int, default=100, help="images for the CPU baseline (0 = skip)")
    ap.add_argument("--parity-images", type=int, default=8, help="images checked against the CPU reference (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="0 = only the headline measurement")
    ap.add_argument("--sym8", type=int, default=-1,
                    help="QCNN_OPT_SYM8 (eight-wave symmetric workgroups): -1 = library default (1 = planner), 0 off, 2 forced tile form, 3 forced sliding form")
    ap.add_argument("--dry-run-shared-gpu", type=int, default=0,
                    help="1 = DRY RUN of the N > 1 code path on a box with fewer GPUs: every rank computes on GPU 0 and the collectives "
                         "(arena broadcast, checksum exchange, barriers, max-over-ranks) run over gloo through host memory.  The line is "
                         "labelled dry_run and its value is NOT a multi-GPU measurement; RCCL itself is not exercised")
    ap.add_argument("--group-capi-child", default="", help=argparse.SUPPRESS)     # internal: path of the parent's top-5 (see group_capi_child)
    ap.add_argument("--group-timeout", type=int, default=300, help="seconds the value_group_capi child process may take")
    ap.add_argument("--init-timeout", type=int, default=120,
                    help="N > 1: seconds the RCCL rendezvous + parameter broadcast + checksum exchange may take before the run "
                         "gives up with a one-line JSON error")
    ap.add_argument("--half8", type=int, default=-1,
                    help="QCNN_OPT_HALF8 (half-panel eight-wave workgroups): -1 = library default (1 = planner), 0 off, 2 forced tile form, 3 forced sliding form")
    ap.add_argument("--streams", type=int, default=1,
                    help="sub-batches of whole panels run concurrently on separate HIP streams (QCNN_OPT_STREAMS; the "
                         "library default is 2).  1 keeps one launch per layer, so that the per-kernel HIP-event "
                         "durations behind `roofline` are those of kernels that own the whole GPU")
    args = ap.parse_args()

    if args.group_capi_child:
        return group_capi_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dry = bool(args.dry_run_shared_gpu) and world > 1
    if dry:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    watchdog = None
    if world > 1:
        # The first N > 1 run is also the first time RCCL initialises with more than one rank on this code: a hang in the
        # rendezvous, the communicator set-up or the parameter broadcast must end as ONE JSON error line, not as the
        # driver's timeout.  The watchdog is disarmed once the broadcast has been verified.
        import datetime
        import threading

        def _give_up():
            print(json.dumps({"metric": "images/sec AlexNet quantized forward", "value": None, "n_gpus": world, "rank": rank,
                              "error": "RCCL set-up (init_process_group / parameter broadcast / checksum exchange) did not finish "
                                       "within %d s on rank %d" % (args.init_timeout, rank)}), flush=True)
            os._exit(3)
        watchdog = threading.Timer(args.init_timeout, _give_up)
        watchdog.daemon = True
        watchdog.start()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.init_timeout))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                    timeout=datetime.timedelta(seconds=args.init_timeout))
    cdev = "cpu" if dry else dev          # where the collectives' tensors live

    topo, synth, capi, dmod, perf = pkg("topology"), pkg("synth"), pkg("capi"), pkg("dist"), pkg("perfmodel")
    in_chw, layers, _, _ = topo.MODELS[args.model]
    sizes = topo.fmap_sizes(in_chw, layers)
    data_root = os.path.join(ROOT, "oracle", "_ref", "data")     # staged DATA files of the reference (not its code): parameters, mean image
    have_shipped = args.model == "AlexNet" and os.path.exists(os.path.join(data_root, "AlexNet/Bin.Files/bvlc_alexnet_aCaF.ctrdLst.01.bin"))
    if args.params == "shipped" and not have_shipped:
        raise SystemExit("bench.py --params shipped: %s holds no staged AlexNet parameter files" % data_root)
    shipped = have_shipped and args.params != "synthetic"
    if shipped:
        params = synth.load_alexnet_shipped(data_root, layers, fixture=1)
        params2 = synth.load_alexnet_shipped(data_root, layers, fixture=2)   # non-degenerate tail: second parity pass only
    else:
        params = synth.make_params(in_chw, layers, seed=0)  # every rank knows the SHAPES; rank 0 owns the VALUES
        params2 = None
    B = args.batch
    strong = world > 1 and args.scaling == "strong"
    lo, hi = dmod.shard_bounds(B, rank, world) if strong else (0, B)
    n_local = hi - lo

    # one explicit stream for everything (torch ops, H2D copies and the engine's kernels): the default stream's
    # handle is NULL, which the C-ABI reads as "create your own stream" — the copies would then not be ordered
    # with the forward passes
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = pkg("engine").QcnnEngine(local, stream.cuda_stream)
    eng.set_option(capi.OPT_LUT_MODE, {"mfma": capi.LUT_MFMA, "exact": capi.LUT_EXACT, "f16": capi.LUT_MFMA_F16}[args.lut])
    eng.set_option(capi.OPT_KEEP_ALL, 0)
    eng.set_option(capi.OPT_PROFILE, 1)
    eng.set_option(capi.OPT_STREAMS, args.streams)
    if args.sym8 >= 0:
        eng.set_option(capi.OPT_SYM8, args.sym8)
    if args.half8 >= 0:
        eng.set_option(capi.OPT_HALF8, args.half8)
    shapes = {i: tuple(int(x) for x in p["ctrd"].shape) for i, p in params.items()}
    eng.configure(in_chw, layers, shapes)
    arena = torch.zeros(eng.arena_bytes(), dtype=torch.uint8, device=dev)
    eng.commit(B, arena.data_ptr())
    if rank == 0:
        eng.upload(params)
    bcast_ms = 0.0
    if world > 1:
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        if dry:
            host = arena.cpu()
            dist.broadcast(host, src=0)
            arena.copy_(host)
            del host
        else:
            dist.broadcast(arena, src=0)                    # RCCL over xGMI: codebooks + row-offset tables + biases
        torch.cuda.synchronize(dev)
        bcast_ms = 1000.0 * (time.perf_counter() - t0)
    if rank != 0:
        eng.mark_loaded()
    rccl_ranks, arena_sums = world, None
    if world > 1:
        # every rank hashes ITS arena on the device; all ranks compare before anything is timed: a broadcast that moved nothing
        # (or the wrong bytes) must not surface as a fast run with wrong class scores on seven shards
        ok_sum, pairs = dmod.checksums_agree(eng.arena_checksum(), device=cdev)
        rccl_ranks = dmod.verified_world_size(device=cdev)       # counted by a real all_reduce over the communicator
        arena_sums = ["%016x:%016x" % p for p in pairs]
        if not ok_sum:
            if rank == 0:
                print(json.dumps({"metric": "images/sec AlexNet quantized forward", "value": None, "n_gpus": world,
                                  "error": "parameter broadcast: the ranks' arenas differ", "arena_checksums": arena_sums}), flush=True)
            raise SystemExit("bench.py: arena checksums differ across ranks after the RCCL broadcast: %r" % (arena_sums,))
        watchdog.cancel()

    # synthetic device-resident input: 8-bit pixels minus the BGR channel means (range of BmpImgIO's output); image i of
    # the global batch is the same whatever the number of ranks
    if shipped:
        # pixel = U{0..255} (numpy default_rng(1234)) minus the centre crop of the shipped mean image
        imgs = torch.from_numpy(synth.make_images(B, in_chw, seed=1234, mean=synth.shipped_mean_image(data_root))).to(dev)
    else:
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        full = torch.randint(0, 256, (B,) + tuple(in_chw), generator=g, device=dev, dtype=torch.int32)
        imgs = full.to(torch.float32)
        del full
        imgs -= torch.tensor([104.0, 117.0, 123.0], device=dev)[: in_chw[0]].view(1, -1, 1, 1)
    classes = sizes[-1][0] * sizes[-1][1] * sizes[-1][2]
    prob = torch.empty((B, classes), dtype=torch.float32, device=dev)
    top5 = torch.empty((B, 5), dtype=torch.int16, device=dev)
    mine = imgs[lo:hi]

    def step():
        if n_local > 0:
            eng.forward_dev(mine.data_ptr(), n_local, prob[lo:hi].data_ptr(), top5[lo:hi].data_ptr())

    def measure(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        eng.reset_layer_ms()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def snapshot():
        """Per-layer HIP-event times of the forwards since the last reset + which kernel family every conv / FC layer ran."""
        ms, rec = eng.layer_ms()
        cf = [i for i, l in enumerate(layers) if l["type"] in (topo.CONV, topo.FCNT)]
        split = {i: eng.layer_split(i) for i in cf}
        dec = {i for i in cf if split[i][0] == -3}                                       # decoded layers
        return dict(layer_ms=ms, recorded=rec,
                    segments={i: eng.layer_segments(i) for i in cf if layers[i]["type"] == topo.CONV},   # sliding kernel, per layer
                    decoded=dec, dec_nchw={i for i in dec if split[i][1] == 2},                          # k_conv_dec_nchw
                    symmetric={i for i in cf if layers[i]["type"] == topo.CONV and split[i][0] == -4},   # k_conv_sym
                    sym8={i for i in cf if split[i][0] in (-5, -6)},                                     # k_conv_sym8 (-6: sliding form), k_fc_sym8
                    half8={i for i in cf if split[i][0] in (-9, -10)})                                   # k_conv_half8 (-10: sliding form)

    dt = measure(step, args.steps, args.warmup)
    snap = snapshot()
    decoded = snap["decoded"]
    ok = bool(torch.isfinite(prob[lo:hi]).all().item())
    images_per_step = B if (strong or world == 1) else B * world

    value_weak = None
    if world > 1 and strong and args.extras:                 # 1000 images per GPU on top of the sharded batch
        eng.set_option(capi.OPT_PROFILE, 0)
        dtw = measure(lambda: eng.forward_dev(imgs.data_ptr(), B, prob.data_ptr(), top5.data_ptr()), max(2, args.steps // 2), 1)
        value_weak = world * B * max(2, args.steps // 2) / dtw
        eng.set_option(capi.OPT_PROFILE, 1)

    extras = {}
    bf_prob = None
    acc_prob = None
    dt_tab, snap_tab = None, None
    if args.extras and rank == 0 and world == 1:
        eng.set_option(capi.OPT_PROFILE, 0)
        if args.streams == 1:
            # the library's default execution mode: two sub-batches on two streams
            eng.set_option(capi.OPT_STREAMS, 2)
            step(); step()
            extras["value_two_streams"] = round(B * 10 / timed(torch, dev, step, 10), 2)
            eng.set_option(capi.OPT_STREAMS, 1)
        if args.model == "AlexNet":
            # every layer through look-up tables + indexed accumulation (QCNN_OPT_DECODE = 0): the north star's algorithm for
            # the two degenerate layers too (conv1: one sub-space of 3 dims; fc8: one-dim sub-spaces), which `value` evaluates
            # through the code words their assignments name
            eng.set_option(capi.OPT_DECODE, 0)
            eng.set_option(capi.OPT_PROFILE, 1)
            dt_tab = measure(step, args.steps, 1)
            snap_tab = snapshot()
            eng.set_option(capi.OPT_PROFILE, 0)
            eng.set_option(capi.OPT_DECODE, 1)
            step()
        # BASELINE configs[4]: fp16 LUT STORAGE (QCNN_OPT_LUT_MODE = 2, opt-in: outside the 1e-4 bar): half-size tables in LDS,
        # fp32 sums — the rate next to the error it costs (against value_tables_only: both run every layer through tables)
        if args.lut == "mfma" and args.model == "AlexNet":
            eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA_F16)
            step()
            extras["value_fp16_lut"] = round(B * 3 / timed(torch, dev, step, 3), 2)
            bf_prob = prob[: args.parity_images].cpu().numpy() if args.parity_images > 0 else None
            f16_split = {str(i): list(eng.layer_split(i)) for i, l in enumerate(layers) if l["type"] in (topo.CONV, topo.FCNT)}
            eng.set_option(capi.OPT_PROFILE, 1)
            eng.reset_layer_ms()
            step(); step()
            f16_ms, _ = eng.layer_ms()
            eng.set_option(capi.OPT_PROFILE, 0)
            extras["fp16_lut"] = dict(layer_ms={"%02d_%s" % (i, topo.TYPE_NAMES[layers[i]["type"]]): round(float(m), 4)
                                                for i, m in enumerate(f16_ms) if m > 0 and layers[i]["type"] in (topo.CONV, topo.FCNT)},
                                      kernels=f16_split,
                                      note="fp16 table entries (round to nearest even), fp32 sums; against the oracle's study mode the kernels agree "
                                           "to 3 - 4e-5 (conv) / 4 - 8e-6 (FC) per layer, not to rounding: the oracle builds an entry with separately "
                                           "rounded multiply and add (src/CaffeEva.cc:1284-1289), the matrix pipe with a fused chain, so a few entries "
                                           "per thousand land on the other side of an fp16 rounding boundary (DESIGN.md, fp16 study); kernels: qcnn_get_layer_split codes "
                                           "(-7: fp16-storage form of the eight-wave kernels, -8: with packed fp16 running sums and twice "
                                           "the tile; others: entries rounded, f32 slots, fp32 sums)")
            # ... and with the running sums in packed fp16 as well (QCNN_OPT_LUT_MODE = 3: twice the tile per wave)
            eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA_F16ACC)
            step()
            extras["value_fp16_lut_fp16_sums"] = round(B * 3 / timed(torch, dev, step, 3), 2)
            acc_prob = prob[: args.parity_images].cpu().numpy() if args.parity_images > 0 else None
            eng.set_option(capi.OPT_PROFILE, 1)
            eng.reset_layer_ms()
            step(); step()
            a_ms, _ = eng.layer_ms()
            eng.set_option(capi.OPT_PROFILE, 0)
            extras["fp16_lut"]["fp16_sums_layer_ms"] = {"%02d_%s" % (i, topo.TYPE_NAMES[layers[i]["type"]]): round(float(m), 4)
                                                        for i, m in enumerate(a_ms) if m > 0 and layers[i]["type"] in (topo.CONV, topo.FCNT)}
            extras["fp16_lut"]["fp16_sums_kernels"] = {str(i): list(eng.layer_split(i)) for i, l in enumerate(layers) if l["type"] in (topo.CONV, topo.FCNT)}
            eng.set_option(capi.OPT_LUT_MODE, capi.LUT_MFMA)
            step()
        # small batches: one image (the reference's own regime) and one 128-image panel (an 8-GPU shard of the batch)
        for nb, reps, key in ((1, 20, "value_b1"), (128, 10, "value_b128")):
            if nb <= B:
                f = lambda nb=nb: eng.forward_dev(imgs.data_ptr(), nb, prob.data_ptr(), top5.data_ptr())
                f()
                extras[key] = round(nb * reps / timed(torch, dev, f, reps), 2)
        # one GPU's share of the 1000-image batch at 8 / 4 / 2 GPUs (what the strong-scaling curve is made of)
        shard = {}
        for nb in (125, 250, 500):
            if nb <= B:
                for ns in (1, 2):
                    eng.set_option(capi.OPT_STREAMS, ns)
                    f = lambda nb=nb: eng.forward_dev(imgs.data_ptr(), nb, prob.data_ptr(), top5.data_ptr())
                    f()
                    t = timed(torch, dev, f, 10) / 10
                    if nb not in shard or t < shard[nb][0]:
                        shard[nb] = (t, ns)
                extras["value_shard_%d" % nb] = round(nb / shard[nb][0], 2)
        eng.set_option(capi.OPT_STREAMS, args.streams)
        if shard:
            t1 = dt / args.steps
            extras["predicted_strong_scaling"] = dict(
                note="PREDICTION from single-GPU shard times (no communication in the data path; the one-time parameter "
                     "broadcast is outside the loop): speed-up over 1 GPU = t(1000 images) / t(1000/G images)",
                speedup={str(B // nb): round(t1 / shard[nb][0], 2) for nb in sorted(shard, reverse=True)},
                shard_ms={str(nb): round(1e3 * shard[nb][0], 4) for nb in sorted(shard)},
                shard_streams={str(nb): shard[nb][1] for nb in sorted(shard)})
        # PCIe-inclusive rates (never `value`).  fp32: qcnn_forward_host_batches — 16 batches from pinned host memory, the
        # upload of batch b + 1 on a copy stream under the layers of batch b, results back through pinned buffers.
        reps = 16        # as the u8 pipeline below: the one upload nothing overlaps is amortised as in a long stream
        pinned = torch.empty(imgs.shape, dtype=torch.float32, pin_memory=True)
        pinned.copy_(imgs)
        host_in = pinned.numpy()
        eng.forward_host_batches([host_in, host_in], want_prob=True, want_top5=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        eng.forward_host_batches([host_in] * reps, want_prob=True, want_top5=True)
        extras["value_incl_pinned_h2d"] = round(B * reps / (time.perf_counter() - t0), 2)
        t0 = time.perf_counter()
        eng.forward_host(host_in)
        extras["value_incl_pinned_h2d_one_batch"] = round(B / (time.perf_counter() - t0), 2)
        del pinned, host_in
        # device-side input pipeline (8-bit 256x256 sources, mean subtraction + crop on the GPU) with the upload of
        # batch i+1 overlapped with the forward pass of batch i: two staging buffers, a copy stream, two events
        hs, ws = max(in_chw[1], 256), max(in_chw[2], 256)
        px = torch.randint(0, 256, (B, in_chw[0], hs, ws), dtype=torch.uint8)
        pinned_u8 = torch.empty(px.shape, dtype=torch.uint8, pin_memory=True)
        pinned_u8.copy_(px)
        stg = [torch.empty(px.shape, dtype=torch.uint8, device=dev) for _ in range(2)]
        mean_img = torch.full((in_chw[0], hs, ws), 110.0, dtype=torch.float32, device=dev)
        copy_stream = torch.cuda.Stream(device=dev)
        copied = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]
        used = [False, False]

        def upload(b):
            with torch.cuda.stream(copy_stream):
                if used[b % 2]:
                    copy_stream.wait_event(freed[b % 2])
                stg[b % 2].copy_(pinned_u8, non_blocking=True)
                copied[b % 2].record(copy_stream)

        def compute(b):
            stream.wait_event(copied[b % 2])
            eng.forward_u8_dev(stg[b % 2].data_ptr(), hs, ws, mean_img.data_ptr(), B, prob.data_ptr(), top5.data_ptr())
            freed[b % 2].record(stream)
            used[b % 2] = True

        def pipeline(k):
            upload(0)
            for b in range(k):
                if b + 1 < k:
                    upload(b + 1)
                compute(b)
        pipeline(2)
        torch.cuda.synchronize(dev)
        k = 16          # the first upload is not overlapped with anything: 16 batches amortise it as a long stream does
        t0 = time.perf_counter()
        pipeline(k)
        torch.cuda.synchronize(dev)
        extras["value_incl_pinned_h2d_u8"] = round(B * k / (time.perf_counter() - t0), 2)
        extras["h2d_note"] = ("f32: qcnn_forward_host_batches, 16 batches from pinned memory, upload of batch b+1 under the "
                              "layers of batch b, probabilities + top-5 returned to the host; _one_batch: a single "
                              "qcnn_forward_host call (two-panel chunks pipelined inside the call); u8: 8-bit sources "
                              "uploaded on a copy stream while the previous batch computes (double buffered)")
        del pinned_u8, stg, px
        # the same batch through the C-ABI's own device group (qcnn_group_*: ONE process, every visible GPU, contiguous image
        # blocks, rank 0's arena broadcast by RCCL inside the library and verified by checksum): device-resident blocks,
        # layers enqueued on every rank's stream, one sync — the single-process figure next to the one-process-per-GPU `value`.
        # In a CHILD process with a time limit: on a node with several GPUs this is RCCL's first multi-rank run from this code,
        # and a hang or crash there must never take the headline measurement down with it.
        step()
        torch.cuda.synchronize(dev)
        ref_top5 = os.path.join(tempfile.gettempdir(), "qcnn_bench_top5_%d.npy" % os.getpid())
        np.save(ref_top5, top5.cpu().numpy())                       # the single context's top-5 of this batch: the child compares
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--group-capi-child", ref_top5, "--batch", str(B),
                                "--params", args.params, "--streams", str(args.streams), "--model", args.model],
                               capture_output=True, text=True, timeout=args.group_timeout)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                g = json.loads(lines[-1])
                extras["value_group_capi"] = g.pop("value")
                extras["group_capi"] = g
            else:
                extras["group_capi"] = dict(error="child exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
        except subprocess.TimeoutExpired:
            extras["group_capi"] = dict(error="no result within %d s (child process killed)" % args.group_timeout)
        except Exception as e:                                       # never the headline's problem
            extras["group_capi"] = dict(error=str(e))
        finally:
            if os.path.exists(ref_top5):
                os.remove(ref_top5)
        torch.cuda.set_device(local)
        eng.set_option(capi.OPT_PROFILE, 1)

    parity = None
    cb = None
    if rank == 0 and (args.parity_images > 0 or (args.cpu_sample > 0 and world == 1)):
        kind, cpu, _ = cpu_side(in_chw, layers, params)
        if args.parity_images > 0 and n_local > 0:
            pn = min(args.parity_images, n_local)
            # re-run the timed configuration on the first images' own batch (same kernels, same mode) and fetch what
            # the reference is compared with: probabilities, top-5 and the last pooling map
            step()
            torch.cuda.synchronize(dev)
            # maps of the fast path that exist on both sides: the last pooling map and the (ReLU-fused) outputs of the
            # hidden FC layers behind it
            fcs = [i for i, l in enumerate(layers) if l["type"] == topo.FCNT]
            fm_ids = [max(i + 1 for i, l in enumerate(layers) if l["type"] == topo.POOL)] + [i + 2 for i in fcs[:-1]]
            # half of the checked images from the first panel, half from the END of this rank's block (the last, ragged panel)
            head = (pn + 1) // 2
            tail = pn - head if n_local >= pn else 0
            idx = list(range(head)) + list(range(n_local - tail, n_local))
            sel = torch.tensor(idx, device=dev)

            def fetch(l):
                return np.concatenate([eng.layer_output_range(l, 0, head)] +
                                      ([eng.layer_output_range(l, n_local - tail, tail)] if tail else []))
            sel_imgs = mine[sel].cpu().numpy()
            parity = parity_check(kind, cpu, layers, sel_imgs, prob[lo:hi][sel].cpu().numpy(),
                                  top5[lo:hi][sel].cpu().numpy().view(np.uint16), {l: fetch(l) for l in fm_ids})
            parity["image_indices"] = [lo + i for i in idx]
            parity["parameters"] = "shipped + fc6 fixture 1" if shipped else "seeded synthetic"
            if shipped:
                # with fixture 1 the real network's fc7 is all-negative (fc8 = bias, one top-5 for every image: SURVEY.md §8c
                # "Trap"): the tail is checked a second time with the fc6 table of fixture 2, same kernels, same images
                fc6 = synth.ALEXNET_FC6
                eng.upload({fc6: params2[fc6]})
                step()
                torch.cuda.synchronize(dev)
                kind2, cpu2, _ = cpu_side(in_chw, layers, params2)
                p2 = parity_check(kind2, cpu2, layers, sel_imgs, prob[lo:hi][sel].cpu().numpy(),
                                  top5[lo:hi][sel].cpu().numpy().view(np.uint16), {l: fetch(l) for l in fm_ids[1:]})
                p2["parameters"] = "shipped + fc6 fixture 2 (non-degenerate tail)"
                parity["tail_fixture2"] = p2
                parity["ok"] = bool(parity["ok"] and p2["ok"])
                eng.upload({fc6: params[fc6]})
                step()
                torch.cuda.synchronize(dev)
            if bf_prob is not None:                      # fp16 table storage: its soft-max outputs against the f32 tables' (same images)
                ref = prob[lo:lo + bf_prob.shape[0]].cpu().numpy()
                extras["fp16_lut"]["max_rel_diff_prob_vs_f32_tables"] = float(
                    max(np.abs(bf_prob[i] - ref[i]).max() / np.abs(ref[i]).max() for i in range(bf_prob.shape[0])))
                if acc_prob is not None:
                    extras["fp16_lut"]["fp16_sums_max_rel_diff_prob_vs_f32_tables"] = float(
                        max(np.abs(acc_prob[i] - ref[i]).max() / np.abs(ref[i]).max() for i in range(acc_prob.shape[0])))
        if args.cpu_sample > 0 and world == 1:
            cb = cpu_baseline(kind, cpu, imgs[: args.cpu_sample].cpu().numpy(), params)

    ref_main = None
    if args.extras and rank == 0 and world == 1 and args.model == "AlexNet" and B <= 1024:
        ref_main = via_reference_main(params, imgs.cpu().numpy())

    vgg = None
    if args.extras and rank == 0 and world == 1 and args.model == "AlexNet":
        # BASELINE.json configs[3]: VGG-16, synthetic parameters in the repo's CaffePara layout, a short run
        eng.close()
        del imgs, prob, top5, arena
        torch.cuda.empty_cache()
        v_chw, v_layers, _, _ = topo.MODELS["VGG16"]
        v_params = synth.make_params(v_chw, v_layers, seed=0)
        v_sizes = topo.fmap_sizes(v_chw, v_layers)
        vb = args.vgg_batch                              # N = 1000 (its feature maps are ~60 GB of the 288 GB)
        ve = pkg("engine").QcnnEngine(local, stream.cuda_stream)
        ve.set_option(capi.OPT_KEEP_ALL, 0)
        ve.set_option(capi.OPT_PROFILE, 1)
        ve.set_option(capi.OPT_STREAMS, 1)
        ve.load_model(v_chw, v_layers, v_params, vb)
        vi = torch.randint(0, 256, (vb,) + tuple(v_chw), device=dev, dtype=torch.uint8).to(torch.float32) - 110.0
        vp = torch.empty((vb, 1000), dtype=torch.float32, device=dev)
        vt = torch.empty((vb, 5), dtype=torch.int16, device=dev)
        vf = lambda: ve.forward_dev(vi.data_ptr(), vb, vp.data_ptr(), vt.data_ptr())
        vf()
        torch.cuda.synchronize(dev)
        ve.reset_layer_ms()
        vdt = timed(torch, dev, vf, 2)
        vms, _ = ve.layer_ms()
        vdom = int(np.argmax(vms))
        fam = lambda code: 8 if code in (-5, -6) else "h8" if code in (-9, -10) else code == -4
        rep = perf.layer_report(v_sizes, v_layers, v_params, vdom, vb, float(vms[vdom]), ve.layer_segments(vdom), fam(ve.layer_split(vdom)[0]))
        conv_total = sum(float(vms[i]) for i, l in enumerate(v_layers) if l["type"] == topo.CONV)
        slow = {}
        for i in sorted((i for i, l in enumerate(v_layers) if l["type"] in (topo.CONV, topo.FCNT)), key=lambda i: -vms[i])[:5]:
            split = ve.layer_split(i)[0]
            r = (perf.decoded_report(v_sizes, v_layers, i, vb, float(vms[i]), ve.layer_split(i)[1] == 2) if (split == -3 and v_layers[i]["type"] == topo.CONV) else
                 perf.layer_report(v_sizes, v_layers, v_params, i, vb, float(vms[i]), ve.layer_segments(i), fam(split)))
            r["ms"] = round(float(vms[i]), 4)
            r["in_hwc"], r["out_hwc"] = list(v_sizes[i]), list(v_sizes[i + 1])
            slow["%02d_%s" % (i, topo.TYPE_NAMES[v_layers[i]["type"]])] = r
        v_lut_flop = sum(perf.conv_work(v_sizes[i], v_sizes[i + 1], l, *[int(x) for x in v_params[i]["ctrd"].shape])["alg_flop"] / 128.0
                         for i, l in enumerate(v_layers) if l["type"] == topo.CONV)
        # parity of THIS configuration (library defaults, batch vb, the kernels timed above): the first and the last image of
        # the batch (first panel / ragged last panel) through the CPU reference — soft-max outputs, top-5, the last pooling
        # map and the hidden FC maps (tests/test_gpu_vgg16_config3.py checks every materialised map of three images)
        v_parity = None
        if args.parity_images > 0:
            vkind, vcpu, _ = cpu_side(v_chw, v_layers, v_params)
            vidx = [0, vb - 1] if vb > 1 else [0]
            vsel = torch.tensor(vidx, device=dev)
            vfcs = [i for i, l in enumerate(v_layers) if l["type"] == topo.FCNT]
            vfm = [max(i + 1 for i, l in enumerate(v_layers) if l["type"] == topo.POOL)] + [i + 2 for i in vfcs[:-1]]
            v_parity = parity_check(vkind, vcpu, v_layers, vi[vsel].cpu().numpy(), vp[vsel].cpu().numpy(),
                                    vt[vsel].cpu().numpy().view(np.uint16),
                                    {l: np.concatenate([ve.layer_output_range(l, i, 1) for i in vidx]) for l in vfm})
            v_parity["image_indices"] = vidx
            v_parity["kernels"] = {str(i): list(ve.layer_split(i)) for i, l in enumerate(v_layers) if l["type"] in (topo.CONV, topo.FCNT)}
            v_parity["kernels_note"] = ("qcnn_get_layer_split codes of the timed forwards: -3 decoded code words, -2 16-wave sliding, "
                                        "-5 / -6 eight-wave tile / sliding form (second number: segments per column), -9 / -10 half-panel eight-wave tile / sliding form")
        vgg = dict(value=round(vb * 2 / vdt, 2), unit="images/s", batch=vb, steps=2, parity=v_parity,
                   outputs_finite=bool(torch.isfinite(vp).all().item()), conv_ms_per_batch=round(conv_total, 3),
                   dominant_layer=vdom, dominant_ms=round(float(vms[vdom]), 4), dominant=rep, slowest_layers=slow,
                   lut_build_algorithmic_tflops=round(v_lut_flop * vb * 2 / vdt / 1e12, 2),
                   lut_build_algorithmic_frac_of_f32_mfma_peak=round(v_lut_flop * vb * 2 / vdt / perf.F32_MFMA_FLOPS, 4),
                   parameters="seeded synthetic, conv Cs=8 K=128, fc Cs=4 K=32, classifier Cs=1 K=16")
        ve.close()

    def roofline_for(snap_, value_, ms_step_, profiled=True):
        """The `roofline` object of one measured configuration (snapshot() of its forwards)."""
        layer_ms, recorded = snap_["layer_ms"], snap_["recorded"]
        decoded, dec_nchw, symmetric, sym8, segments = (snap_[k] for k in ("decoded", "dec_nchw", "symmetric", "sym8", "segments"))
        half8 = snap_["half8"]
        dom = int(np.argmax(layer_ms))
        dom_ms = float(layer_ms[dom])
        # a layer is launched once per sub-batch (QCNN_OPT_STREAMS): layer_ms is the mean duration of ONE launch,
        # so the algorithmic bytes / look-ups are those of one launch (its share of the panels)
        panels = (n_local + 127) // 128
        ns = max(1, min(args.streams, panels))
        launch_images = n_local / ns
        abytes = algorithmic_bytes(sizes, layers, params, dom, launch_images)
        achieved = abytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        conv_idx = [i for i, l in enumerate(layers) if l["type"] == topo.CONV]
        name = "%s%d" % (topo.TYPE_NAMES[layers[dom]["type"]], (conv_idx.index(dom) + 1) if dom in conv_idx else dom)
        traffic, tsrc, ttable = (pmc_traffic(dom, ns) if (n_local == 1000 and args.model == "AlexNet" and profiled)
                                 else (None, "batch / model / options differ from the profiled run (the committed PMC passes are "
                                             "of the default configuration)", {}))
        per_layer = {}
        for i, l in enumerate(layers):
            if l["type"] in (topo.CONV, topo.FCNT) and layer_ms[i] > 0:
                r = (perf.decoded_report(sizes, layers, i, launch_images, float(layer_ms[i]), i in dec_nchw) if (i in decoded and l["type"] == topo.CONV) else
                     dict(tile="decoded code words: x @ w on the matrix pipe, 64 channels x 64 images per workgroup",
                          issued_mfma_flop_per_image=2 * sizes[i][0] * sizes[i][1] * sizes[i][2] * ((l["nod"] + 63) // 64 * 64),
                          lookups_replaced_per_image=sizes[i][0] * sizes[i][1] * sizes[i][2] * l["nod"]) if i in decoded else
                     perf.layer_report(sizes, layers, params, i, launch_images, float(layer_ms[i]), segments.get(i), 8 if i in sym8 else "h8" if i in half8 else (i in symmetric)))
                r["ms"] = round(float(layer_ms[i]), 4)
                per_layer["%02d_%s" % (i, topo.TYPE_NAMES[l["type"]])] = r
        total_lk, table_lk = 0, 0                       # all look-ups / those of the layers that really ran as table look-ups
        for i, l in enumerate(layers):
            lk = 0
            if l["type"] == topo.CONV:
                lk = perf.conv_work(sizes[i], sizes[i + 1], l, *shapes[i])["lookups"]
            elif l["type"] == topo.FCNT:
                lk = shapes[i][0] * l["nod"]
            total_lk += lk
            if i not in decoded:
                table_lk += lk
        step_bytes = sum(algorithmic_bytes(sizes, layers, params, l, n_local) for l in range(len(layers))
                         if layer_ms[l] > 0)
        # The pipe that bounds the dominant kernel.  Table kernels (LUT build + indexed accumulation): neither HBM (a few
        # per cent of 8 TB/s) nor the matrix pipe but the LDS read path of the look-ups — `achieved` = look-up bytes (4 B per
        # look-up and image) per second against 256 CUs x 256 B/clk x 2.4 GHz; the HBM and matrix-pipe figures sit beside
        # it.  A decoded layer (products on the matrix pipe): issued f32 MFMA FLOP against the dense f32 peak.
        dom_rep = per_layer.get("%02d_%s" % (dom, topo.TYPE_NAMES[layers[dom]["type"]]), {})
        lds_peak_gbs = perf.LDS_READ_BYTES_PER_CLK * perf.CUS * perf.CLOCK_HZ / 1e9
        if dom in decoded:
            bound = dict(bound="mfma", achieved=round(dom_rep.get("mfma_util", 0.0) * perf.F32_MFMA_FLOPS / 1e12, 2),
                         peak=round(perf.F32_MFMA_FLOPS / 1e12, 1), unit="TFLOP/s", frac=dom_rep.get("mfma_util"))
        else:
            bound = dict(bound="lds", achieved=dom_rep.get("lookup_gbs"), peak=round(lds_peak_gbs, 1), unit="GB/s",
                         frac=dom_rep.get("lds_frac"),
                         mfma_issued_frac=dom_rep.get("mfma_util"), mfma_algorithmic_frac=dom_rep.get("mfma_algorithmic_frac"))
        other = {}
        for l2, ent in sorted(ttable.items()):          # HBM traffic of the other profiled kernels (e.g. the decoded first layer)
            if l2 != dom and layer_ms[l2] > 0:
                ab = algorithmic_bytes(sizes, layers, params, l2, launch_images)
                other[str(l2)] = dict(kernel=ent.get("kernel"), traffic=int(ent["bytes"]), fetched=ent.get("fetch_bytes"),
                                      written=ent.get("write_bytes"), algorithmic_bytes_per_launch=int(ab),
                                      traffic_over_algorithmic=round(ent["bytes"] / ab, 2),
                                      rocprof_avg_ms=ent.get("rocprof_avg_ms"))
        # rocprofv3 --kernel-trace --stats average of the same kernel from the committed profile of THIS kernel source
        # (profiles/*/kernel_stats.csv via traffic.json; hash-checked like `traffic`): the live HIP-event time must agree
        dom_ent = ttable.get(dom, {})
        rp_ms = dom_ent.get("rocprof_avg_ms")
        roof = dict(bound, kernel=("k_conv_dec (layer %d, %s)" % (dom, name)) if dom in decoded else
                    ("k_conv_sym8 (layer %d, %s)" % (dom, name)) if dom in sym8 else
                    ("k_conv_half8 (layer %d, %s)" % (dom, name)) if dom in half8 else
                    ("k_conv_sym (layer %d, %s)" % (dom, name)) if dom in symmetric else
                    "k_%s_aprx (layer %d, %s)" % ("conv" if dom in conv_idx else "fc", dom, name),
                    traffic=traffic, traffic_source=tsrc,
                    traffic_over_algorithmic=round(traffic / abytes, 2) if traffic else None, other_kernels_traffic=other,
                    hbm_achieved=round(achieved, 2), hbm_peak=HBM_PEAK_GBS, hbm_frac=round(achieved / HBM_PEAK_GBS, 5),
                    ms_per_launch=round(dom_ms, 4), launches_timed=recorded * ns, launches_per_step=ns,
                    rocprof_avg_ms_per_launch=rp_ms, rocprof_kernel=dom_ent.get("kernel") if rp_ms else None,
                    rocprof_over_hip_events=round(rp_ms / dom_ms, 3) if (rp_ms and dom_ms > 0) else None,
                    images_per_launch=launch_images, algorithmic_bytes_per_launch=int(abytes),
                    whole_step_hbm_frac=round(step_bytes / (ms_step_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    whole_step_lds_frac=round(table_lk * 4 * value_ / 1e9 / lds_peak_gbs, 4),
                    lds_read_peak="256 CUs x 256 B/clk x 2.4 GHz (ds_read_b64/b128)",
                    layers=per_layer,
                    layer_ms={"%02d_%s" % (i, topo.TYPE_NAMES[layers[i]["type"]]): round(float(m), 4)
                              for i, m in enumerate(layer_ms) if m > 0})
        return roof, total_lk, table_lk, ns

    if rank == 0:
        ms_step = 1000.0 * dt / args.steps
        value = images_per_step * args.steps / dt
        roof, total_lk, table_lk, ns = roofline_for(snap, value, ms_step)
        conv_fc = [i for i, l in enumerate(layers) if l["type"] in (topo.CONV, topo.FCNT)]
        lname = lambda i: "%s%d" % ("conv" if layers[i]["type"] == topo.CONV else "fc", conv_fc.index(i) + 1)
        dec_names = [lname(i) for i in sorted(decoded)]
        scheme = ("fp32 LUT + uint8 indices for %s; %s — one sub-space of 3 dims / one-dim sub-spaces — evaluate the same sums as "
                  "f32 MFMA products of the code words their uint8 assignments name (QCNN_OPT_DECODE, library default); "
                  "every layer through LUTs: value_tables_only" % (", ".join(lname(i) for i in conv_fc if i not in decoded),
                                                                   " and ".join(dec_names))) if decoded else "fp32 LUT + uint8 indices, every conv / FC layer"
        par = ("one %d-image batch sharded over %d GPU(s) in contiguous blocks, parameters replicated by one RCCL "
               "broadcast" % (B, world)) if (strong or world == 1) else (
            "%d images per GPU on %d GPUs, parameters replicated by one RCCL broadcast" % (B, world))
        out = {
            "metric": "images/sec AlexNet quantized forward" if args.model == "AlexNet" else "images/sec %s quantized forward" % args.model,
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # the north star's algorithm — look-up tables + uint8-indexed accumulation — on EVERY conv / FC layer (QCNN_OPT_DECODE = 0;
            # `value` runs conv1 and fc8 through the code words their assignments name): filled in below when measured
            "alg_north_star_value": None, "alg_north_star_ms_per_step": None, "alg_north_star_roofline_frac": None,
            "alg_north_star_note": None,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s Q-CNN approximate forward (%s), one batch of %d synthetic %dx%d "
                                   "images per step, device-resident" % (args.model, scheme, images_per_step, in_chw[1], in_chw[2]),
                       "decoded_layers": dec_names,
                       "global_batch": images_per_step, "images_on_rank0": n_local, "lut_builder": args.lut,
                       "parameters": ("the reference's shipped AlexNet files (oracle/_ref/data) + fc6 assignments of SURVEY.md §8c "
                                      "fixture 1 (the mount lacks that file); inputs U{0..255} minus the shipped mean image") if shipped
                       else "seeded synthetic (seed 0), shipped AlexNet quantisation shapes",
                       "streams_per_gpu": ns, "parallelism": par},
            "rccl_ranks": 0 if dry else rccl_ranks, "param_broadcast_ms": round(bcast_ms, 3),
            **({"dry_run": "N > 1 code path on ONE GPU shared by %d ranks, collectives over gloo through host memory: NOT a multi-GPU "
                           "measurement (`value` is meaningless); RCCL not exercised" % rccl_ranks} if dry else {}),
            "param_broadcast_verified": ("every rank's device arena checksum equals rank 0's: %s" % arena_sums[0]) if arena_sums else
                                        "one rank: nothing to broadcast",
            "outputs_finite": ok,
            "lookups_per_image": int(total_lk),
            "lookups_per_s": round(table_lk * value, 0),
            "lookups_performed_per_image": int(table_lk),
            "lookups_note": ("lookups_per_s counts the %d look-ups per image that ran as table look-ups; layers %s evaluate "
                             "theirs (%d) as products of the code words the assignments name (QCNN_OPT_DECODE)"
                             % (table_lk, sorted(decoded), total_lk - table_lk)) if decoded else "all evaluated as table look-ups",
            "roofline": roof,
        }
        if snap_tab is not None:
            # the north star's scheme for ALL eight conv / FC layers (QCNN_OPT_DECODE = 0): its own rate and roofline block
            v_tab = images_per_step * args.steps / dt_tab
            r_tab, _, lk_tab, _ = roofline_for(snap_tab, v_tab, 1000.0 * dt_tab / args.steps, profiled=False)
            out["alg_north_star_value"] = round(v_tab, 2)
            out["alg_north_star_ms_per_step"] = round(1000.0 * dt_tab / args.steps, 4)
            out["alg_north_star_roofline_frac"] = r_tab.get("frac")
            out["alg_north_star_note"] = ("images/s with LUT build + uint8-indexed accumulation on all eight conv / FC layers (= value_tables_only; "
                                          "its dominant kernel and roofline: roofline_tables_only, bound %s, kernel %s); `value` is the library default, "
                                          "which evaluates conv1 / fc8 — one 3-dim sub-space / one-dim sub-spaces — as f32 MFMA products of the same "
                                          "code words" % (r_tab.get("bound"), r_tab.get("kernel")))
            out["value_tables_only"] = round(v_tab, 2)
            out["ms_per_step_tables_only"] = round(1000.0 * dt_tab / args.steps, 4)
            out["lookups_per_s_tables_only"] = round(lk_tab * v_tab, 0)
            out["roofline_tables_only"] = r_tab
        if out["alg_north_star_value"] is None:
            if not decoded:
                out.update(alg_north_star_value=out["value"], alg_north_star_ms_per_step=out["ms_per_step"], alg_north_star_roofline_frac=roof.get("frac"),
                           alg_north_star_note="`value` itself: no layer of this run was decoded")
            else:
                out["alg_north_star_note"] = "not measured in this run (--extras 0 or N > 1): see value_tables_only of the N = 1 line"
        out.update(extras)
        if value_weak is not None:
            out["value_weak"] = round(value_weak, 2)
        if parity is not None:
            out["parity"] = parity
        if ref_main is not None:
            out["value_via_reference_main"] = ref_main.get("value")
            out["via_reference_main"] = ref_main
        if vgg is not None:
            out["value_vgg16"] = vgg["value"]
            out["vgg16"] = vgg
        if cb is not None:
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
            if "all_cores" in cb:
                out["speedup_vs_cpu_all_cores"] = round(value / cb["all_cores"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                      # rank 0 may still have been busy with the CPU reference
        dist.destroy_process_group()
    if rank == 0 and (not ok or (parity is not None and not parity["ok"]) or
                      (vgg is not None and vgg.get("parity") is not None and not vgg["parity"]["ok"])):
        raise SystemExit("bench.py: outputs differ from the CPU reference beyond %g (see `parity` / `vgg16.parity`)" % TOL)


if __name__ == "__main__":
    main()
