import torch, sys
sys.path.insert(0,'.')
from neko_amd import ops
from oracle import neko_oracle as O
B,T,H,hd=2,333,2,64
d=H*hd
g=torch.Generator().manual_seed(1)
qkv=torch.randn(B,T,3*d,generator=g).to(torch.bfloat16)
mask=torch.ones(B,T)
kb,ks=ops.mask_bias(mask.cuda())
out,lse=ops.attn_fwd(qkv.view(B*T,3*d).cuda().contiguous(),kb,ks,B,T,H,hd)
o=out.float().cpu().view(B,T,H,hd)
nan=torch.isnan(o)
print('nan count',int(nan.sum()), 'of', o.numel())
rows=nan.any(-1).any(-1)  # B,T
for b in range(B):
    idx=rows[b].nonzero().flatten().tolist()
    print('b',b,'nan rows', idx[:10], '...', idx[-5:], len(idx))
hh=nan.any(-1).any(1) ; print('per (b,h) any nan', hh.tolist())
cols=nan.any(0).any(0).any(0); print('hd cols with nan', cols.nonzero().flatten().tolist()[:70])
print('lse nan', int(torch.isnan(lse).sum()))
