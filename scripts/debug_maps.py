import torch, sys
sys.path.insert(0, "/root/repo")
from markovflow_amd import _lib
_lib.load()
print("current stream handle:", torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l or "magma" in l})
print("\n".join(libs))
