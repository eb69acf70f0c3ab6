import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn as nn
from mulactseg_amd import ops
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for cin, cout, k, hw in ((128,64,1,192),(64,256,1,192),(256,64,1,192),(128,256,1,192),(256,128,1,192),(128,512,1,96),(512,128,1,96),(64,64,3,192),(64,64,3,384),(64,128,3,384),(256,1024,1,48),(1024,256,1,48)):
    conv = nn.Conv2d(cin, cout, k, padding=k//2, bias=False).cuda()
    x = torch.randn(4, cin, hw, hw, device='cuda')
    w = conv.weight.detach()
    pf = ops.conv_sk_pack(w, 1, False)
    t_sk = timeit(lambda: ops.conv_sk(x, w, 1, 1, packed=pf))
    t_sks = timeit(lambda: ops.conv_sk(x, w, 1, 1, packed=pf, stats=True))
    with torch.no_grad():
        t_mf = timeit(lambda: ops.conv_mfma(conv, x)) if ops.conv_mfma_supported(conv, x) else float('nan')
    gf = 2.0*4*hw*hw*cin*cout*k*k/1e9
    print("%4d->%4d k%d %3dx%-3d %6.2f GF | conv_sk %6.1f us (%5.1f TF) with stats %6.1f | conv_mfma %6.1f us (%5.1f TF)" % (cin,cout,k,hw,hw,gf,t_sk,gf/t_sk*1e3,t_sks,t_mf,gf/t_mf*1e3))
