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
    """Against (a) the reference's update rule restated with torch ops on the same device -- moments within 3 ulp (ATen may contract
    a * b + c into an fma, csrc/optim.hip never does), parameters within 2 ulp of the UPDATE plus one of the parameter -- and (b)
    today's torch.optim.AdamW (its first moment is a lerp, another rounding of the same number)."""
    dev = _gpu()
    from mulactseg_amd.utils.optim import FusedAdamW
    a, b, c = _params(dev, 1), _params(dev, 1), [q.detach().clone() for q in _params(dev, 1)]
    mk = lambda ps: [{'params': ps[:5], 'lr': 2e-5}, {'params': ps[5:], 'lr': 2e-4}]
    own = FusedAdamW(mk(a), lr=2e-5, weight_decay=1e-5)
    ref = torch.optim.AdamW(mk(b), lr=2e-5, weight_decay=1e-5, foreach=False, fused=False)
    cm, cv = [torch.zeros_like(q) for q in c], [torch.zeros_like(q) for q in c]
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
            # 3 ulp of the LARGER of the two terms of m b1 + g (1 - b1) (they may cancel), relative for the sum of squares
            bound_m = 3.6e-7 * torch.maximum(cm[i].abs(), 0.2 * grads[i].abs()) + 1e-37
            assert bool(((sa['exp_avg'] - cm[i]).abs() <= bound_m + 3.6e-7 * (sa['exp_avg'] - 0.1 * grads[i]).abs()).all()), (step, i)
            rel_v = float(((sa['exp_avg_sq'] - cv[i]).abs() / cv[i].abs().clamp_min(1e-37)).max())
            assert rel_v <= 3.6e-7, (step, i, rel_v)
            tol = 2.4e-7 * lr * 4 + 2 ** -23 * float(c[i].abs().max())          # |update| <= ~lr per step here (|m / sqrt(v)| of order 1)
            assert float((pa.detach() - c[i]).abs().max()) <= tol, (step, i, float((pa.detach() - c[i]).abs().max()), tol)
            assert float((pa.detach() - pb.detach()).abs().max()) <= 8 * tol, (step, i)
            assert torch.allclose(sa['exp_avg'], ref.state[pb]['exp_avg'], rtol=2e-6, atol=1e-30)
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
