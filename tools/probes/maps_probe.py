import sys, os
sys.path[:0]=["knp-emi-fenics-x_amd","examples/idealized_geometries"]
import torch
torch.cuda.init()
from knpemi import _lib
L=_lib.load()
libs=set()
for ln in open("/proc/self/maps"):
    p=ln.split()[-1]
    if any(k in p for k in ("amdhip","hsa-runtime","libhiprtc","rccl","libknpemi")): libs.add(p)
print("\n".join(sorted(libs)))
print({k:v for k,v in os.environ.items() if any(s in k for s in ("HSA","HIP","GPU","AMD","ROC"))})
