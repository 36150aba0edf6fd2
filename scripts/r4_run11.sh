cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
Y5_LIB_PATH=yolov5_amd/libyolov5_hip_x.so python - <<'PY'
import ctypes as C, torch, sys
sys.path.insert(0,'.')
from yolov5_amd import _lib
from yolov5_amd.packing import pack_conv_weight
lib=_lib.lib(); dev=torch.device("cuda:0"); st=C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for name,H,C1 in (("6.b.cv2 128@40",40,128),("13.b.cv2 128@40 nores",40,128)):
    B=64; x=torch.randn((B,H,H,C1),device=dev,dtype=torch.float16); w=torch.randn((C1,C1,3,3),device=dev)*0.05
    wp,bp,K,Kpad,Npad=pack_conv_weight(w,torch.zeros(C1,device=dev),torch.float16); y=torch.zeros((B,H,H,C1),device=dev,dtype=torch.float16)
    ref=None
    for cfg in (76,77,76,77):
        d=_lib.ConvDesc(dtype=_lib.Y5_F16,B=B,H=H,W=H,C1=C1,ldx=C1,OH=H,OW=H,C2=C1,ldy=C1,KH=3,KW=3,SH=1,SW=1,PH=1,PW=1,act=1,Kpad=Kpad,Npad=Npad,ldr=0,ld2=0,cfg=cfg,max_blocks=0)
        ms=C.c_float(0); best=1e9
        for _ in range(3):
            rc=lib.y5_conv2d_time(C.byref(d),C.c_void_p(x.data_ptr()),C.c_void_p(wp.data_ptr()),C.c_void_p(bp.data_ptr()),None,C.c_void_p(y.data_ptr()),None,20,st,C.byref(ms)); best=min(best,ms.value*1e3)
        torch.cuda.synchronize()
        if ref is None: ref=y.clone()
        print(name,'cfg',cfg,f'{best:.1f} us','rc',rc,'maxdiff',float((y.float()-ref.float()).abs().max()))
PY
