"""What F.scaled_dot_product_attention launches around its kernel for the Jacobi forward's shapes (strided K/V views of the
static cache, an additive mask whose last dimension is not a multiple of 16)."""
import torch, torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
dev="cuda"
P,H,S_max,D=64,4,4096,128
G,T=7,40
for S in (450, 448):
    k=torch.randn(P,H,S_max,D,device=dev,dtype=torch.bfloat16); v=torch.randn_like(k)
    q=torch.randn(P,H,G*T,D,device=dev,dtype=torch.bfloat16)
    Sp=(S+15)//16*16
    cases = {"bias [..,S] contiguous": torch.zeros(P,1,G*T,S,device=dev,dtype=torch.bfloat16),
             "bias view [..,:S] of [..,Sp]": torch.zeros(P,1,G*T,Sp,device=dev,dtype=torch.bfloat16)[..., :S],
             "bias expanded over heads from [P,1,1?]": None}
    for name,b in cases.items():
        if b is None: continue
        kk,vv=k[:,:,:S],v[:,:,:S]
        for _ in range(3): F.scaled_dot_product_attention(q,kk,vv,attn_mask=b)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5): F.scaled_dot_product_attention(q,kk,vv,attn_mask=b)
            torch.cuda.synchronize()
        print(f"== S={S} {name}")
        for e in prof.key_averages():
            if e.device_time_total>0: print(f"   {e.key[:80]:80s} calls {e.count:3d} avg {e.device_time_total/e.count:8.1f} us")
