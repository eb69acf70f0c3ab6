import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from mulactseg_amd.models import get_model, deeplab
dev = torch.device('cuda:0')
rep = {}
for size in (96, 128):
    torch.manual_seed(21)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
    x = torch.randn(2, 3, size, size, generator=torch.Generator(device=dev).manual_seed(8), device=dev)
    deeplab.path_report(reset=True)
    z = net(x, lowres=True)
    z.square().mean().backward()
    rep[size] = deeplab.path_report(reset=True)
print(json.dumps(rep[96], indent=0)); print(json.dumps(rep[128], indent=0))
# per-module forward determinism at 96: hook outputs over two runs
torch.manual_seed(21)
net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
x = torch.randn(2, 3, 96, 96, generator=torch.Generator(device=dev).manual_seed(8), device=dev)
runs = []
for r in range(2):
    outs = {}
    hs = []
    for n, m in net.named_modules():
        if n:
            hs.append(m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, (o[0] if isinstance(o, (tuple, list)) else o).detach().clone() if torch.is_tensor(o) or isinstance(o, (tuple, list)) and torch.is_tensor(o[0]) else None)))
    net(x, lowres=True)
    for h in hs: h.remove()
    runs.append(outs)
first = [n for n in runs[0] if runs[0][n] is not None and n in runs[1] and not torch.equal(runs[0][n], runs[1][n])]
print('modules whose output differs between two forwards (in call order):', first[:12])
