"""Stage a data root with this repo's synthetic AlexNet parameters + images and run the reference's unmodified Main.cc
(build/bin/QuanCNN_hip) on it with the environment given on the command line (KEY=VALUE ...)."""
import importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n: importlib.import_module("quantized-cnn_amd." + n)
topo, synth, fileio = pkg("topology"), pkg("synth"), pkg("fileio")
in_chw, layers, _, _ = topo.MODELS["AlexNet"]
params = synth.make_params(in_chw, layers, seed=0)
n = 1000
imgs = synth.make_images(n, in_chw, seed=3)
with tempfile.TemporaryDirectory() as root:
    os.makedirs(os.path.join(root, "AlexNet", "Bin.Files")); os.makedirs(os.path.join(root, "ILSVRC12.227x227.IMG"))
    synth.write_param_dir(os.path.join(root, "AlexNet", "Bin.Files"), "bvlc_alexnet_aCaF", params)
    fileio.write_bin(os.path.join(root, "ILSVRC12.227x227.IMG", "dataMatTst.single.bin"), imgs)
    fileio.write_bin(os.path.join(root, "ILSVRC12.227x227.IMG", "lablVecTst.uint16.bin"), np.zeros((1, 1, 1, n), np.uint16))
    for spec in sys.argv[1:] or [""]:
        env = dict(os.environ, QCNN_BATCH=str(n), QCNN_BATCHES="8", QCNN_DEVICE="0")
        env.update(kv.split("=", 1) for kv in spec.split(",") if kv)
        r = subprocess.run([os.path.join(ROOT, "build", "bin", "QuanCNN_hip")], cwd=root, capture_output=True, text=True, env=env)
        keep = [l for l in (r.stdout + r.stderr).splitlines() if "swDebugTimePri" in l or "swAllLayers" in l or "pipeline]" in l or "elapsed" in l or "ERROR" in l]
        print("==", spec or "(default)"); print("\n".join(keep))
