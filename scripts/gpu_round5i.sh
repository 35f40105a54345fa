# round 5: the rocprofv3 passes of the headline (profiles/r5_v11) and of VGG-16 (profiles/r5_vgg16)
bash scripts/gpu_prof.sh > gpurun_out/prof.log 2>&1
tail -5 gpurun_out/prof.log
bash scripts/gpu_prof_vgg.sh > gpurun_out/prof_vgg.log 2>&1
tail -8 gpurun_out/prof_vgg.log | cut -c1-300
