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

    def kernel_prefix(tile):
        """bench `roofline.layers[*].tile` -> how the kernel's name starts in the counter tables"""
        import re
        m = re.match(r"half panels, 8 waves (\d+)x(\d+)x(\d+) \((\d+) wave set", tile)       # k_conv_half8<cpw, th, tw, ws, ..>
        if m:
            ws = int(m.group(4))
            return "k_conv_half8<%d.%s.%s.%d." % (int(m.group(3)) * ws // 8, m.group(1), m.group(2), ws)
        m = re.match(r"half panels, slide 8 waves (\d+) column\(s\) x (\d+) slots x (\d+)", tile)
        if m:
            ch, tw = int(m.group(3)), int(m.group(1))
            ws = 2 if ch in (128, 192) else 1
            return "k_conv_half8<%d.%s.%d.%d." % (ch * ws // 8, m.group(2), tw, ws)
        m = re.match(r"symmetric 8 waves (\d+)x(\d+)x(\d+)", tile)
        if m:
            return "k_conv_sym8<%d.%s.%s." % (int(m.group(3)) // 8, m.group(1), m.group(2))
        if tile.startswith("symmetric"):
            return "k_conv_sym<"
        m = re.match(r"slide (\d+) column\(s\) x (\d+) slots x (\d+)", tile)
        if m:
            return "k_conv_aprx<%s.%s.%d.8." % (m.group(1), m.group(2), int(m.group(3)) // 12)
        if tile.startswith("decoded code words: x @ w"):
            return "k_fc_dec"
        if tile.startswith("decoded code words, NCHW"):
            return "k_conv_dec_nchw"
        if tile.startswith("decoded"):
            return "k_conv_dec<"
        m = re.match(r"(\d+)x(\d+)x(\d+)$", tile)
        if m:
            return "k_conv_aprx<%s.%s.%d.8." % (m.group(1), m.group(2), int(m.group(3)) // 12)
        return None

    # rocprofv3 --kernel-trace --stats: average duration per kernel, names normalised like the counter tables' ("k_x<1.2.false>")
    import re as _re
    stats = {}
    for r in rows[1:]:
        m = _re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r[0])
        if m:
            stats[(m.group(1) + (m.group(2) or "")).replace(", ", ".")] = (float(r[3]) / 1e6, int(r[1]))   # (average ms, calls)
    kernels = {}
    for key, rep in bench["roofline"]["layers"].items():
        pre = kernel_prefix(rep.get("tile", ""))
        rows_ = [r for r in table if pre and r["kernel"].startswith(pre) and r.get("FETCH_SIZE") and r.get("WRITE_SIZE")]
        if key.endswith("_conv") and len(rows_) >= 1:
            r = max(rows_, key=lambda r: float(r["SQ_WAVE_CYCLES"] or 0))
            f, w = float(r["FETCH_SIZE"]), float(r["WRITE_SIZE"])
            avg = stats.get(r["kernel"])
            kernels[str(int(key[:2]))] = dict(kernel=r["kernel"], bytes=int((2.0 * f + w) * 1024), fetch_bytes=int(2.0 * f * 1024),
                                              write_bytes=int(w * 1024), rocprof_avg_ms=round(avg[0], 4) if avg else None,
                                              rocprof_calls=avg[1] if avg else None, hip_event_ms_under_rocprof=rep.get("ms"))
    dom = kernels[str(dom_layer)]
    fetch_kib, write_kib = dom["fetch_bytes"] / 2048.0, dom["write_bytes"] / 1024.0
    dom = dict(kernel=dom["kernel"])
    sys.path.insert(0, ROOT)
    import bench as bench_mod
    h = bench_mod.kernel_hash()          # every device source under quantized-cnn_amd/csrc
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
