#!/usr/bin/env python3
"""Register / scratch report of every gfx950 kernel in the built device objects (quantized-cnn_amd/csrc/*.hip.o).

Reads the code-object metadata the compiler wrote (AMDGPU notes: .vgpr_count, .sgpr_count, .vgpr_spill_count,
.sgpr_spill_count, .private_segment_fixed_size, .group_segment_fixed_size) — no GPU needed.  Every hot kernel sits at
128 VGPRs (16 waves x 128 = the unified register file of a CU), so a spill is a design error, not noise:
tests/test_isa_report.py fails when a kernel a shipped model launches needs scratch.

usage: scripts/isa_report.py [--csv]            (build first: python -c "import __graft_entry__ as g; g.build()")
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quantized-cnn_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def kernels_of(obj_path):
    """[(demangled name, {field: int})] of one host object with an embedded gfx950 code object."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj_path))
        shutil.copyfile(obj_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in glob.glob(local + ".*") if "gfx950" in f]
        if not dev:
            return out
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev[0]], check=True, capture_output=True, text=True).stdout
    # one metadata entry per kernel, its keys in alphabetical order: ".args" opens an entry, ".name" sits in the middle of it
    cur = None
    for line in notes.splitlines():
        if re.match(r"  - \.\w+:", line):             # a new kernel of `amdhsa.kernels` (argument lists are indented deeper)
            cur = {}
            out.append(cur)
        m = re.match(r"\s+(?:-\s+)?\.(\w+):\s+(\S+)", line)
        if not m or cur is None:
            continue
        key, val = m.group(1), m.group(2)
        if key == "name" and "name" not in cur:
            cur["name"] = val
        elif key in FIELDS and key not in cur:
            cur[key] = int(val)
    out = [[d.pop("name"), d] for d in out if "name" in d]
    if out:
        names = subprocess.run(["c++filt"] + [n for n, _ in out], check=True, capture_output=True, text=True).stdout.splitlines()
        for ent, nm in zip(out, names):
            ent[0] = re.sub(r"\(anonymous namespace\)::", "", nm)
    return [(n, d) for n, d in out]


def short(name):
    """k_conv_aprx<1, 3, 12, 8, 2, true>(ConvParams, ...) -> k_conv_aprx<1,3,12,8,2,true>"""
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name


def report():
    rows = []
    for obj in sorted(glob.glob(os.path.join(CSRC, "*.hip.o"))):
        for name, d in kernels_of(obj):
            rows.append((os.path.basename(obj)[:-6], short(name), d))
    return rows


def main():
    rows = report()
    print("file,kernel," + ",".join(FIELDS))
    for f, k, d in rows:
        print("%s,\"%s\",%s" % (f, k, ",".join(str(d.get(x, "")) for x in FIELDS)))
    bad = [(f, k, d) for f, k, d in rows if d.get("private_segment_fixed_size", 0) or d.get("vgpr_spill_count", 0) or d.get("sgpr_spill_count", 0)]
    print("# %d kernels, %d with spills / scratch" % (len(rows), len(bad)), file=sys.stderr)
    for f, k, d in bad:
        print("#   %s %s: vgpr_spill=%s sgpr_spill=%s scratch=%s B" % (f, k, d.get("vgpr_spill_count"), d.get("sgpr_spill_count"),
                                                                      d.get("private_segment_fixed_size")), file=sys.stderr)


if __name__ == "__main__":
    main()
