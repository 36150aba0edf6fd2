"""Kernel-experiment aid (GPU): y5_sppf_pool_bwd on the yolov5s bs=64 P5 shape; Y5_SPPF_BWD_GV = 1 | 2 | 4 selects the channel groups per workgroup."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from yolov5_amd import _lib
lib=_lib.lib(); dev=torch.device("cuda:0")
B,H,Cc,k=64,20,256,5
act=torch.randn((B,H,H,4*Cc),device=dev).half(); grad=torch.randn((B,H,H,4*Cc),device=dev).half()
st=C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def run(): return lib.y5_sppf_pool_bwd(C.c_void_p(act.data_ptr()),C.c_void_p(grad.data_ptr()),B,H,H,Cc,4*Cc,4*Cc,k,st)
assert run()==0
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("Y5_SPPF_BWD_GV","default(2)"), "sppf_pool_bwd us", e0.elapsed_time(e1)/20*1e3)
