import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulactseg_amd.models import get_model
def run(tag, cl, bench):
    torch.backends.cudnn.benchmark = bench
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).cuda().train()
    x = torch.randn(4, 3, 768, 768, device='cuda')
    if cl:
        net = net.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-5)
    def step():
        opt.zero_grad(set_to_none=True)
        y = net(x)
        (y.float().mean()).backward()
        opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize()
    print(tag, "ms/step", (time.perf_counter()-t0)/6*1e3, flush=True)
for tag, cl, b in (("nchw", False, False), ("nchw+benchmark", False, True), ("channels_last", True, False), ("channels_last+benchmark", True, True)):
    try: run(tag, cl, b)
    except Exception as e: print(tag, "failed", repr(e)[:200])
