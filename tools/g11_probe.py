"""G11 (tests/golden/g11_train.npz) through this package's modules as plain PyTorch ops on the host at 1 and 8 threads: how far the
reference's own f32 arithmetic moves when only its summation order changes (the yardstick of tests/test_train_golden.py:G11_SELF)."""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, time
import test_train_golden as T
from oracle import port
g = np.load(T.GOLDEN11)
x, tgt, spx, msk = T._inputs(g)
for th in (1, 8):
    torch.set_num_threads(th)
    t0=time.time()
    net, opt, sched = T._build(g, 'cpu')
    xt, tt, ts, tm = torch.from_numpy(x), torch.from_numpy(tgt), torch.from_numpy(spx), torch.from_numpy(msk)
    S, Tm = int(g['S']), float(g['temp'])
    q = {}
    net.classifier.register_forward_hook(lambda m, i, o: q.__setitem__('q', o.detach()))
    preds = net(xt)
    group = port.group_max_ce(preds, tt, ts, tm, S, Tm, 'onlymulti')
    ce, mc = port.merged_positive_ce(preds, tt, ts, tm, Tm, 'decomp')
    loss = 16.0 * ce + 8.0 * mc + 1.0 * group
    loss.backward()
    d, rel = T._logit_dev(g, 1, q['q'].numpy(), preds.detach().numpy())
    got = np.array([float(loss.detach()), float(ce.detach()), float(mc.detach()), float(group.detach())])
    l = np.abs(got / g['losses1'].astype(np.float64) - 1).max()
    # grads
    num=den=0; worst=0
    per = []
    for i,(n,p) in enumerate(net.named_parameters()):
        ref = g['grad_%03d'%i]; gotc = T.sub256(p.grad.numpy())
        a = float(((gotc.astype(np.float64)-ref)**2).sum()); b=float((ref.astype(np.float64)**2).sum())
        num+=a; den+=b; per.append(((a/max(b,1e-300))**.5, n))
        gn = float(p.grad.double().norm()); worst=max(worst, abs(gn-float(g['gnorm_%03d'%i]))/max(float(g['gnorm_%03d'%i]),1e-30))
    per.sort(reverse=True)
    print('threads', th, 'logits', d, rel, 'losses rel', l, 'grad relL2', (num/den)**.5, 'gnorm worst', worst, 'worst layers', per[:4], 'sec', time.time()-t0, flush=True)
