import torch, time
print(hasattr(torch.ops.aten, 'miopen_convolution_relu'), hasattr(torch.ops.aten, 'miopen_convolution_add_relu'))
dev='cuda'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
import sys; sys.path.insert(0,'.')
from mulactseg_amd import ops
import torch.nn as nn
shapes=[(4,64,256,512,64,1,1,0),(4,64,256,512,64,3,1,1),(4,256,256,512,64,1,1,0),(4,128,128,256,128,3,1,1),(4,512,128,256,128,1,1,0),(4,256,64,128,256,3,1,1),(4,1024,64,128,256,1,1,0),(4,512,64,128,512,3,1,2)]
for (N,Ci,H,W,Co,k,s,p) in shapes:
    x=torch.randn(N,Ci,H,W,device=dev); w=torch.randn(Co,Ci,k,k,device=dev)*0.05; b=torch.randn(Co,device=dev)
    bn=nn.BatchNorm2d(Co).to(dev).eval()
    dil = 2 if p==2 else 1
    with torch.no_grad():
        a=t(lambda: ops.bn_act(bn, torch.nn.functional.conv2d(x,w,None,s,p,dil), True))
        try:
            f=t(lambda: torch.ops.aten.miopen_convolution_relu(x,w,b,[s,s],[p,p],[dil,dil],1))
        except Exception as e:
            f=float('nan'); print('err',str(e)[:100])
        c=t(lambda: torch.nn.functional.conv2d(x,w,None,s,p,dil))
    print((N,Ci,H,W,Co,k), 'conv+bn_act %.3f ms  fused miopen %.3f ms  conv only %.3f'%(a,f,c))
