import sys; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from latent2im_amd import conv
T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
for (cin,cout,k,s,p,h,b) in [(64,64,3,1,1,1,4),(32,64,1,1,0,2,4),(16,32,3,2,1,2,4),(256,512,1,2,0,2,4),(64,64,3,1,1,2,4),(2048,512,1,1,0,1,4),(512,2048,1,1,0,1,4)]:
    rs=np.random.RandomState(cin+h)
    w=T(rs.randn(cout,cin,k,k)/np.sqrt(cin*k*k)); x=T(rs.randn(b,cin,h,h))
    ref=F.conv2d(x,w,stride=s,padding=p)
    fc=conv.FrozenConv2d(w,s,p,device='cuda')
    for hint in (0,4,5,2):
        try:
            y=fc.forward(x.cuda(), tile_hint=hint).cpu()
            err=[(y[i]-ref[i]).abs().max().item() for i in range(b)]
            print((cin,cout,k,s,p,h,b),'hint',hint,'per-sample err',['%.1e'%e for e in err])
        except Exception as e: print('hint',hint,'ERR',str(e)[:80])
