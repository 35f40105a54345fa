#!/usr/bin/env python3
"""Turn the scratch output of scripts/gpu_prof.sh (gpurun_out/prof) into a tracked profiles/<name>/ directory:
kernel_stats.csv (rocprofv3 --kernel-trace --stats, our kernels only), pmc_summary.csv (mean per dispatch of every
counter pass), traffic.json (HBM bytes of the dominant kernel: 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, with the
hash of the kernel source they were measured on — bench.py only quotes them for that source), bench_under_rocprof.json.
usage: make_profile_dir.py profiles/r2_v7 [gpurun_out/prof]"""
import csv
import hashlib
import io
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    dst = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof")
    os.makedirs(dst, exist_ok=True)
    rows = list(csv.reader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "(anonymous namespace)::k_" in r[0]:
                w.writerow(r)
    pmc = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), src], text=True)
    open(os.path.join(dst, "pmc_summary.csv"), "w").write(pmc)
    shutil.copyfile(os.path.join(src, "trace_bench.json"), os.path.join(dst, "bench_under_rocprof.json"))
    table = list(csv.DictReader(io.StringIO(pmc)))
    bench = json.loads(open(os.path.join(src, "trace_bench.json")).readline())
    dom_layer = int(bench["roofline"]["kernel"].split("layer ")[1].split(",")[0])
    # the table-kernel family (tile / sliding / symmetric conv, FC): the dominant one by wave cycles is the layer the bench names
    dom = max((r for r in table if r["kernel"].startswith(("k_conv_aprx", "k_conv_sym", "k_fc_aprx"))),
              key=lambda r: float(r["SQ_WAVE_CYCLES"] or 0))
    fetch_kib, write_kib = float(dom["FETCH_SIZE"]), float(dom["WRITE_SIZE"])
    sys.path.insert(0, ROOT)
    import bench as bench_mod
    h = bench_mod.kernel_hash()          # every device source under quantized-cnn_amd/csrc
    # per-layer table: the dominant table kernel and the decoded first layer (k_conv_dec runs layer 0 only)
    kernels = {str(dom_layer): dict(kernel=dom["kernel"], bytes=int((2.0 * fetch_kib + write_kib) * 1024),
                                    fetch_bytes=int(2.0 * fetch_kib * 1024), write_bytes=int(write_kib * 1024))}
    for r in table:
        if r["kernel"].startswith("k_conv_dec") and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
            f, w = float(r["FETCH_SIZE"]), float(r["WRITE_SIZE"])
            kernels.setdefault("0", dict(kernel=r["kernel"], bytes=int((2.0 * f + w) * 1024), fetch_bytes=int(2.0 * f * 1024),
                                         write_bytes=int(w * 1024)))
    json.dump({"kernel": dom["kernel"], "layer": dom_layer, "launches_per_forward": 1, "kernels": kernels,
               "bytes": int((2.0 * fetch_kib + write_kib) * 1024),
               "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib, "kernel_hash": h,
               "note": "rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes "
                       "read for 4-, 8- and 16-byte-per-lane streams alike and WRITE_SIZE the exact bytes (calibration.txt: "
                       "scripts/ubench/copy_calib.hip, 2 GiB per direction), so bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024."},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "traffic.json")).read())


if __name__ == "__main__":
    main()
