"""Every forced kernel family reproducible run to run and equal to the tile kernels — a short form of
scripts/soak_modes.py (the race hunt a builder runs with 40 repetitions) inside the driver-run GPU tier: the same 300-image
AlexNet batch and a 140-image VGG-16 batch (two panels, a ragged one) through the 16-wave tile / sliding / symmetric kernels,
the eight-wave tile and sliding forms and the planner's own choice; every forward of a mode must reproduce its first one bit
for bit, and all modes must agree with the tile kernels (bitwise for the conv families; to 1e-5 where the eight-wave FC kernel
groups its partial sums differently).  Same sums as src/CaffeEva.cc:816-865 in the same (kh, kw, m) order whatever the family."""
import importlib.util
import os

import pytest

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu


def _soak():
    spec = importlib.util.spec_from_file_location("soak_modes", os.path.join(ROOT, "scripts", "soak_modes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("model,n,reps", [("AlexNet", 300, 3), ("VGG16", 140, 2)])
def test_forced_kernel_families_reproducible_and_equal(model, n, reps):
    capi = pkg("capi")
    modes = [("tile", []), ("slide16", [(capi.OPT_SLIDE, 2)]), ("sym16", [(capi.OPT_SYM, 2)]),
             ("sym8 tile", [(capi.OPT_SYM8, 2)]), ("sym8 slide", [(capi.OPT_SYM8, 3)]),
             ("half8 tile", [(capi.OPT_HALF8, 2)]), ("half8 slide", [(capi.OPT_HALF8, 3)]),
             ("planner", [(capi.OPT_SLIDE, 1), (capi.OPT_SYM, 1), (capi.OPT_SYM8, 1), (capi.OPT_HALF8, 1)])]
    _soak().run(model, n, reps, modes)          # asserts inside
