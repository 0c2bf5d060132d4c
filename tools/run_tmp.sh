cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path[:0]=['tests','.','neutts-air_amd']
import ctypes as C, torch, numpy as np
from neutts import _hip
lib=_hip.load_library()
def bf(x): return x.to(torch.bfloat16)
for (M,N,K,variant) in [(300,272,384,4),(256,768,2048,1),(256,1280,768,2),(64,64,128,2),(64,64,4096,2)]:
    g=torch.Generator().manual_seed(M*1000+N+K)
    xq=(torch.randn(M,K,generator=g)*4).clamp(-448,448).to(torch.float8_e4m3fn)
    wq=(torch.randn(N,K,generator=g)*32).clamp(-448,448).to(torch.float8_e4m3fn)
    ws=torch.ones(N); xs=1.0
    acc64=(xq.double()@wq.double().t())
    acc32=(xq.float()@wq.float().t())
    out=torch.empty(M,N,dtype=torch.bfloat16,device='cuda')
    # use scale 2^-12 so results are bf16-friendly but keep exactness of scaling
    ws=ws*2.0**-12
    xd,wd,sd=xq.view(torch.uint8).cuda(),wq.view(torch.uint8).cuda(),ws.cuda()
    assert lib.ntts_k_gemm_fp8(C.c_void_p(xd.data_ptr()),C.c_void_p(wd.data_ptr()),C.c_void_p(sd.data_ptr()),xs,None,C.c_void_p(out.data_ptr()),M,N,K,variant)==0
    got=out.float().cpu().double()*2.0**12
    ref=acc64
    ulp=2.0**(torch.floor(torch.log2(ref.abs().clamp(min=1e-30)))-7)
    e_gpu=((got-ref).abs()/ulp); e_t32=((bf(acc32.float()).double()-ref).abs()/ulp)
    print(f"{M}x{N}x{K} v{variant}: GPU vs fp64 ref (bf16 ulps) mean {e_gpu.mean():.3f} max {e_gpu.max():.2f} frac>1 {float((e_gpu>1).float().mean()):.4f} | torch fp32->bf16 mean {e_t32.mean():.3f} max {e_t32.max():.2f} | rel acc err GPU rms {float(((got-ref)/ref.abs().clamp(min=1)).pow(2).mean().sqrt()):.2e}")
PY
