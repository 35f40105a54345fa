"""quantized-cnn_amd — MI355X-native Quantized-CNN approximate forward pass (LUT build + indexed
accumulation), behind the reference's CaffeEva / CaffePara interface.

The directory name carries a hyphen (it mirrors the reference repo's name), so import it with
``importlib.import_module("quantized-cnn_amd")`` after putting the repo root on ``sys.path``;
``bench.py`` / ``tests/conftest.py`` / ``__graft_entry__.py`` do exactly that.

Sub-modules: ``fileio`` (.bin/.cbn formats), ``topology`` (layer tables), ``synth`` (seeded parameter
sets), ``build`` (hipcc / g++ recipes), ``capi`` (ctypes binding of include/qcnn_hip.h),
``engine`` (CaffeEva-shaped Python driver used by bench.py and the tests).
"""
__all__ = ["fileio", "topology", "synth", "build", "capi", "engine"]
