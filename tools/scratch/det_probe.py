import sys, os
sys.path.insert(0, os.getcwd())
import torch
from mulactseg_amd.models import get_model, deeplab
dev = torch.device('cuda:0')
for size, N in ((96, 2), (128, 2), (256, 2), (768, 4)):
    for drop in (0.0, None):
        torch.manual_seed(21)
        net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
        if drop is not None:
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = drop
        x = torch.randn(N, 3, size, size, generator=torch.Generator(device=dev).manual_seed(8), device=dev)
        outs = []
        for rep in range(3):
            torch.manual_seed(5)
            for p in net.parameters():
                p.grad = None
            deeplab.path_report(reset=True)
            z = net(x, lowres=True)
            z.square().mean().backward()
            outs.append((z.detach().clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters()])))
        print(size, N, 'dropout', drop, 'logits equal', [bool(torch.equal(outs[0][0], o[0])) for o in outs[1:]],
              'maxdiff', [float((outs[0][0] - o[0]).abs().max()) for o in outs[1:]],
              'grads equal', [bool(torch.equal(outs[0][1], o[1])) for o in outs[1:]], flush=True)
