"""Register / scratch budget of the built gfx950 kernels (CPU tier: reads the code-object metadata, no GPU).

Every hot kernel is written against exactly 128 VGPRs per wave (16 waves x 128 = the unified register file of a CU), or
256 for the 8-wave symmetric kernels: every design decision there is a register decision, so a spill to scratch in a
kernel that a shipped model launches in its default configuration is a defect, not noise ("Spills")."""
import importlib.util
import os
import re

import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location("isa_report", os.path.join(ROOT, "scripts", "isa_report.py"))
isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa)

# kernels the shipped topologies launch with the default options at the headline batch (AlexNet, 1000 images: decoded
# conv1 / fc8, symmetric conv2, tile conv3 / conv4, sliding conv5, FC, fused LRN + pool) and at one panel / VGG-16
HOT = [
    r"k_conv_dec<6,1,false,2,4>", r"k_conv_dec<3,1,false,3,4>", r"k_conv_dec<4,1,true,2,4>", r"k_conv_dec_nchw<6,4>",
    r"k_conv_sym<2>", r"k_conv_sym8<.*>", r"k_fc_sym8<.*>", r"k_conv_half8<.*>",
    r"k_conv_aprx<1,1,32,8,2,false>", r"k_conv_aprx<1,1,24,8,2,false>", r"k_conv_aprx<1,2,16,8,2,false>",
    r"k_conv_aprx<1,3,12,8,2,false>", r"k_conv_aprx<2,2,8,8,[12],false>", r"k_conv_aprx<2,3,6,8,2,false>",
    r"k_conv_aprx<1,3,12,8,2,true>", r"k_conv_aprx<1,2,16,8,2,true>", r"k_conv_aprx<1,3,8,8,1,true>",
    r"k_fc_aprx<32,2,1>", r"k_fc_dec", r"k_lrn_pool<5,true>", r"k_lrn_stream.*", r"k_pool4.*", r"k_pack.*",
    r"k_softmax_lds.*", r"k_top5_lds.*", r"k_conv_sum", r"k_sum_partials",
]


@pytest.fixture(scope="module")
def rows():
    r = isa.report()
    if not r:
        pytest.skip("device objects not built (python -c 'import __graft_entry__ as g; g.build()')")
    return r


def test_hot_kernels_need_no_scratch(rows):
    seen = set()
    for _, name, d in rows:
        for pat in HOT:
            if re.fullmatch(pat, name):
                seen.add(pat)
                assert d.get("private_segment_fixed_size", 0) == 0 and d.get("vgpr_spill_count", 0) == 0, \
                    "%s spills: %s" % (name, d)
    missing = [p for p in HOT if p not in seen]
    assert not missing, "hot kernels not found in the build: %s" % missing


def test_register_budgets(rows):
    """16-wave workgroups (1024 threads) fit 128 VGPRs, 8-wave ones 256: more would not launch one workgroup per CU."""
    for _, name, d in rows:
        limit = {1024: 128, 512: 256}.get(d.get("max_flat_workgroup_size", 0))
        if limit and name.startswith(("k_conv_", "k_fc_")):
            assert d.get("vgpr_count", 0) + d.get("agpr_count", 0) <= limit, (name, d)


def test_generated_gather_header_is_current():
    """quantized-cnn_amd/csrc/qcnn_sym8_gather.h is what scripts/gen/gen_sym8_gather.py (2 reads per block, 3 register sets) emits."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "scripts", "gen", "gen_sym8_gather.py"), "2", "3"], text=True)
    assert out == open(os.path.join(root, "quantized-cnn_amd", "csrc", "qcnn_sym8_gather.h")).read()
