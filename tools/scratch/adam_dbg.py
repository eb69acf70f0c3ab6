import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import test_optim_gpu as T
from mulactseg_amd.utils.optim import FusedAdamW
dev = torch.device('cuda:0')
a = T._params(dev, 1)
c = [q.detach().clone() for q in T._params(dev, 1)]
own = FusedAdamW([{'params': a[:5], 'lr': 2e-5}, {'params': a[5:], 'lr': 2e-4}], lr=2e-5, weight_decay=1e-5)
cm, cv = [torch.zeros_like(q) for q in c], [torch.zeros_like(q) for q in c]
g = torch.Generator(device='cpu').manual_seed(7)
for step in range(3):
    grads = []
    for pa in a:
        gr = (torch.randn(pa.shape, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))).to(dev)
        pa.grad = gr.clone(); grads.append(gr)
    own.step()
    for i in range(len(a)):
        lr = 2e-5 if i < 5 else 2e-4
        mp = cm[i].clone()
        T._torch_1_11_step(c[i], grads[i], cm[i], cv[i], step + 1, lr)
        m = own.state[a[i]]['exp_avg']
        d = (m - cm[i]).abs()
        k = int(d.reshape(-1).argmax())
        # the same update with explicit separate roundings on the device
        alt = (mp * 0.9) + (grads[i] * torch.tensor(0.1, device=dev))
        print(step, i, 'max|dm|', float(d.max()), 'at m_prev', float(mp.reshape(-1)[k]), 'g', float(grads[i].reshape(-1)[k]), 'own', float(m.reshape(-1)[k]),
              'aten', float(cm[i].reshape(-1)[k]), 'separate', float(alt.reshape(-1)[k]), 'own==separate', bool(torch.equal(m, alt)))
