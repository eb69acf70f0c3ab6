"""utils/optim.FusedAdamW (csrc/optim.hip) against torch's single-tensor AdamW -- the update rule of the reference's optimizer
(trainer/base.py:64-66) -- on the same parameters and gradients: several steps, two parameter groups with moving learning rates,
odd sizes and unaligned views, the device-side skip flag, the state dictionary."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device('cuda:0')


def _params(dev, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    shapes = [(64, 3, 3, 3), (257,), (128, 64, 1, 1), (1,), (20, 256), (4099,), (2048, 512, 1, 1), (7, 5, 3, 3)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


def _torch_1_11_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5):
    """torch 1.11 ``_single_tensor_adamw`` (the reference's environment, actsegmul.yml:95-99) statement by statement, in place."""
    import math
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def test_matches_the_single_tensor_update_of_torch_over_several_steps():
    """(a) The kernel's moments equal its documented arithmetic bit for bit (separate roundings, from its own previous moments);
    (b) against the reference's update rule restated with torch ops on the same device, and against today's torch.optim.AdamW:
    parameters within 2 ulp of the UPDATE plus one of the parameter (observed: bit-equal over five steps), moments within the
    few ulp that ATen's fma contraction / lerp leave in a moving average."""
    dev = _gpu()
    from mulactseg_amd.utils.optim import FusedAdamW
    a, b, c = _params(dev, 1), _params(dev, 1), [q.detach().clone() for q in _params(dev, 1)]
    mk = lambda ps: [{'params': ps[:5], 'lr': 2e-5}, {'params': ps[5:], 'lr': 2e-4}]
    own = FusedAdamW(mk(a), lr=2e-5, weight_decay=1e-5)
    ref = torch.optim.AdamW(mk(b), lr=2e-5, weight_decay=1e-5, foreach=False, fused=False)
    cm, cv = [torch.zeros_like(q) for q in c], [torch.zeros_like(q) for q in c]
    prev_m, prev_v = [torch.zeros_like(q) for q in c], [torch.zeros_like(q) for q in c]
    gmax = [0.0] * len(c)
    g = torch.Generator(device='cpu').manual_seed(7)
    for step in range(5):
        grads = []
        for pa, pb in zip(a, b):
            gr = (torch.randn(pa.shape, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))).to(dev)
            pa.grad, pb.grad = gr.clone(), gr.clone()
            grads.append(gr)
        lrs = [(2e-5 if k == 0 else 2e-4) * (1 - step / 10) ** 0.9 for k in range(2)]
        for o in (own, ref):                                         # the poly schedule moves both groups every step
            for k, grp in enumerate(o.param_groups):
                grp['lr'] = lrs[k]
        own.step()
        ref.step()
        for i, (pa, pb) in enumerate(zip(a, b)):
            lr = lrs[0 if i < 5 else 1]
            _torch_1_11_step(c[i], grads[i], cm[i], cv[i], step + 1, lr)
            sa = own.state[pa]
            # the kernel's own arithmetic, exactly: every product and sum rounded separately, from ITS previous moments
            m_sep = prev_m[i] * 0.9 + grads[i] * torch.tensor(0.1, device=dev)
            v_sep = prev_v[i] * 0.999 + torch.tensor(0.001, device=dev) * (grads[i] * grads[i])
            assert torch.equal(sa['exp_avg'], m_sep) and torch.equal(sa['exp_avg_sq'], v_sep), (step, i)
            prev_m[i], prev_v[i] = sa['exp_avg'].clone(), sa['exp_avg_sq'].clone()
            # against ATen's sequence (which contracts a * b + c into an fma: one rounding less per step, and the difference rides
            # along in the moving averages): a few ulp of the largest gradient the tensor has seen; parameters: observed bit-equal
            gmax[i] = max(gmax[i], float(grads[i].abs().max()))
            d_m = float((sa['exp_avg'] - cm[i]).abs().max())
            rel_v = float(((sa['exp_avg_sq'] - cv[i]).abs() / cv[i].abs().clamp_min(1e-37)).max())
            tol = 2.4e-7 * lr * 8 + 2 ** -22 * float(c[i].abs().max())          # |update| <= ~lr per step here (|m / sqrt(v)| of order 1)
            d_p = float((pa.detach() - c[i]).abs().max())
            d_ref = float((pa.detach() - pb.detach()).abs().max())
            report = dict(step=step, i=i, d_m=d_m, gmax=gmax[i], rel_v=rel_v, d_p=d_p, tol=tol, d_ref=d_ref)
            assert d_m <= 1e-6 * gmax[i] and rel_v <= 2e-6 and d_p <= tol and d_ref <= 8 * tol, report
    assert float(own.state[a[0]]['step']) == 5.0 and float(ref.state[b[0]]['step']) == 5.0


def test_skip_flag_on_the_device_leaves_everything_untouched_and_state_round_trips():
    dev = _gpu()
    from mulactseg_amd.utils.optim import FusedAdamW
    ps = _params(dev, 3)
    opt = FusedAdamW([{'params': ps[:4]}, {'params': ps[4:], 'lr': 1e-2}], lr=1e-3, weight_decay=1e-2)
    assert all(g['fused'] for g in opt.param_groups)                 # (trainer/base.py:guard_optimizer_step reads this)
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    before = [p.detach().clone() for p in ps]
    m_before = [opt.state[p]['exp_avg'].clone() for p in ps]
    opt.found_inf = torch.ones((), device=dev)
    opt.step()
    assert all(torch.equal(p.detach(), q) for p, q in zip(ps, before))
    assert all(torch.equal(opt.state[p]['exp_avg'], m) for p, m in zip(ps, m_before)) and float(opt.state[ps[0]]['step']) == 1.0
    opt.found_inf = torch.zeros((), device=dev)
    opt.step()
    assert all(not torch.equal(p.detach(), q) for p, q in zip(ps, before)) and float(opt.state[ps[0]]['step']) == 2.0
    del opt.found_inf
    # state dictionary: torch's layout, loadable by torch.optim.AdamW and back
    import copy
    sd = copy.deepcopy(opt.state_dict())            # (state_dict() hands out the state tensors themselves; load_state_dict may keep them)
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'} and len(sd['state']) == len(ps)
    qs = _params(dev, 3)
    for q, p in zip(qs, ps):
        q.data.copy_(p.data)
    twin = torch.optim.AdamW([{'params': qs[:4]}, {'params': qs[4:], 'lr': 1e-2}], lr=1e-3, weight_decay=1e-2, foreach=False, fused=False)
    twin.load_state_dict(sd)
    again = FusedAdamW([{'params': qs[:4]}, {'params': qs[4:], 'lr': 1e-2}], lr=1e-3, weight_decay=1e-2)
    again.load_state_dict(copy.deepcopy(twin.state_dict()))
    for p, q in zip(ps, qs):
        p.grad = torch.full_like(p, 0.5)
        q.grad = torch.full_like(q, 0.5)
    opt.step()
    again.step()
    assert all(torch.equal(p.detach(), q.detach()) for p, q in zip(ps, qs)) and float(again.state[qs[0]]['step']) == 3.0


def test_weights_move_under_the_packed_image_cache():
    """The optimizer step hook that invalidates the packed weight images (ops._bump_param_epoch) fires for this optimizer too."""
    dev = _gpu()
    from mulactseg_amd import ops
    from mulactseg_amd.utils.optim import FusedAdamW
    p = torch.nn.Parameter(torch.randn(8, 8, device=dev))
    opt = FusedAdamW([p], lr=1e-2)
    e0 = ops._PARAM_EPOCH[0]
    p.grad = torch.ones_like(p)
    opt.step()
    assert ops._PARAM_EPOCH[0] == e0 + 1
