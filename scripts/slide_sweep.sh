mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x -k "sliding or vgg or other_reference" 2>&1 | tail -3
for sl in 1; do python - <<PY 2>&1 | grep -vE "^layerInd|^\[INFO\]|^\[CHECK|amdgpu.ids" | tee -a gpurun_out/slide_sweep.log
import importlib, sys, time, torch, numpy as np
sys.path.insert(0, '.')
pkg = lambda n: importlib.import_module('quantized-cnn_amd.' + n)
capi, topo, synth = pkg('capi'), pkg('topology'), pkg('synth')
in_chw, layers, _, _ = topo.MODELS['VGG16']
params = synth.make_params(in_chw, layers, seed=0)
eng = pkg('engine').QcnnEngine(0)
eng.set_option(capi.OPT_KEEP_ALL, 0); eng.set_option(capi.OPT_STREAMS, 1); eng.set_option(capi.OPT_SLIDE, $sl); eng.set_option(capi.OPT_PROFILE, 1)
eng.load_model(in_chw, layers, params, 256)
x = torch.randint(0, 256, (256,) + tuple(in_chw), device='cuda', dtype=torch.int32).to(torch.float32) - 110.0
t5 = torch.empty((256, 5), dtype=torch.int16, device='cuda')
eng.forward_dev(x.data_ptr(), 256, None, t5.data_ptr()); eng.sync(); eng.reset_layer_ms()
t0 = time.perf_counter()
for _ in range(2): eng.forward_dev(x.data_ptr(), 256, None, t5.data_ptr())
eng.sync(); dt = (time.perf_counter() - t0) / 2
ms, _ = eng.layer_ms()
conv = [i for i, l in enumerate(layers) if l['type'] == topo.CONV]
print('VGG16 slide $sl: %.1f img/s; conv ms %s; cuts %s' % (256 / dt, ' '.join('%.2f' % ms[i] for i in conv), ' '.join('%dx%d' % eng.layer_split(i) for i in conv)))
PY
done
