cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path[:0]=['tests','.','neutts-air_amd']
import ctypes as C, torch, numpy as np
from neutts import _hip
lib=_hip.load_library()
def run(xq,wq,scale=1.0):
    M,K=xq.shape; N=wq.shape[0]
    ws=torch.full((N,),scale); out=torch.empty(M,N,dtype=torch.bfloat16,device='cuda')
    xd,wd,sd=xq.to(torch.float8_e4m3fn).view(torch.uint8).cuda(),wq.to(torch.float8_e4m3fn).view(torch.uint8).cuda(),ws.cuda()
    assert lib.ntts_k_gemm_fp8(C.c_void_p(xd.data_ptr()),C.c_void_p(wd.data_ptr()),C.c_void_p(sd.data_ptr()),1.0,None,C.c_void_p(out.data_ptr()),M,N,K,2)==0
    return out.float().cpu()
g=torch.Generator().manual_seed(1)
# 1) small integers: every partial sum exact -> must match exactly
x=torch.randint(-2,3,(64,256),generator=g).float(); w=torch.randint(-2,3,(64,256),generator=g).float()
ref=(x@w.t()); got=run(x,w)
print("small ints: mismatches", int((got!=ref.to(torch.bfloat16).float()).sum()), "of", ref.numel())
# 2) one big product + many small ones, per k position of the big one
for small in (1.0, 0.125, 2.0**-6):
    res=[]
    for pos in (0,5,31,32,100,127):
        x=torch.full((16,128),1.0); w=torch.full((64,128),small); x[:,pos]=448.0; w[:,pos]=448.0
        got=run(x,w,scale=2.0**-10)[0,0].item()*2**10
        exact=448.0*448.0+127*small
        res.append((pos,got,exact))
    print("big+small", small, [(p, g_, e) for p,g_,e in res])
# 3) cancellation: +big -big + small
x=torch.zeros(16,128); w=torch.zeros(64,128); x[:,0]=448; w[:,0]=448; x[:,1]=448; w[:,1]=-448; x[:,2:]=1.0; w[:,2:]=1.0
print("cancel: got", run(x,w)[0,0].item(), "exact", 126.0)
x=torch.zeros(16,128); w=torch.zeros(64,128); x[:,0]=448; w[:,0]=448; x[:,64]=448; w[:,64]=-448; x[:,2:64]=1.0; w[:,2:64]=1.0
print("cancel across the two MFMAs of a chunk / k-steps: got", run(x,w)[0,0].item(), "exact", 62.0)
PY
