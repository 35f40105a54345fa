import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib,subprocess,os,sys
b=importlib.import_module('quantized-cnn_amd.build')
name=sys.argv[1]; flags=sys.argv[2:]
so=os.path.join(b.PKG,'libqcnn_hip_%s.so'%name)
srcs=[os.path.join(b.CSRC,s) for s in b.HIP_SOURCES]
subprocess.check_call([b.HIPCC]+b.HIP_FLAGS+flags+['-shared','-o',so]+srcs+['-L/opt/rocm/lib','-lrccl','-lpthread'], stderr=subprocess.DEVNULL)
print(so)
