import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from transformerbasednavierstokesolver_amd import ops
B,N,heads,D,M = 2,4096,8,32,64
C=heads*D
rng=np.random.default_rng(1)
xf=torch.from_numpy(rng.standard_normal((B,N,2*C)).astype(np.float32)).cuda()
ws=torch.from_numpy((rng.standard_normal((M,D))*D**-0.5).astype(np.float32)).cuda()
bs=torch.from_numpy((0.3*rng.standard_normal(M)).astype(np.float32)).cuda()
temp=torch.tensor([0.03,0.5,7.0,0.25,1.5,0.1,5.0,0.8]).cuda()
spart,npart=ops.slice_scatter(xf,2*C,0,xf,2*C,C,ws,bs,temp,B,N,heads,D,M)
nch=npart.shape[1]; ppc=N//nch
x=xf[...,:C].double().view(B,N,heads,D).permute(0,2,1,3)
tau=temp.double().clamp(0.1,5.0).view(1,heads,1,1)
w=torch.softmax((x@ws.double().t()+bs.double())/tau,-1)   # B,h,N,M
ref=w.view(B,heads,nch,ppc,M).sum(3).view(B*heads,nch,M)
err=(npart.double()-ref)
print("nchunk",nch,"max abs err",err.abs().max().item(), "rel", (err.norm()/ref.norm()).item())
e=err.abs()
idx=(e>1e-4).nonzero()
print("bad count",idx.shape[0],"of",e.numel())
print(idx[:40].tolist())
print("per-mt err", e.view(B*heads,nch,4,16).amax((0,1,3)).tolist())
print("per-head err", e.view(B,heads,nch,M).amax((0,2,3)).tolist())
print("per-li err", e.view(B*heads,nch,4,16).amax((0,1,2)).tolist())
