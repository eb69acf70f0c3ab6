"""Test-time stand-in for the un-vendored dependency ``torch_scatter`` (pytorch-scatter 2.0.9,
pinned by the reference's ``actsegmul.yml:99``).  ORACLE / TEST INFRASTRUCTURE ONLY.

It exists so that the reference's own Python (``/root/reference``) can be imported *in the build
container* to generate the golden vectors under ``tests/golden/`` (see ``oracle/gen_golden.py``).
Nothing in ``mulactseg_amd/`` imports it and it never runs on the GPU box.

The semantics restated here are the published CPU algorithm of pytorch-scatter 2.0.9:

* ``scatter(..., reduce='sum')``  : ``zeros(dim_size).scatter_add_(dim, index, src)`` -- on CPU a
  sequential accumulation in source order.
* ``scatter(..., reduce='mean')`` : the sum above divided by a count that is itself a scatter-sum
  of ones **in src.dtype**, clamped to >= 1; true division for floating types, floor division for
  integer types.
* ``scatter(..., reduce='max')`` / ``scatter_max`` : returns ``(out, arg)``; destination rows that
  received nothing are 0 in ``out`` and ``src.size(dim)`` in ``arg``; ties keep the FIRST source
  index (strict ``>`` update in source order); the backward routes the gradient to ``arg`` only.

parity unpinned at this boundary: the reference holds no test for it; this
file is short on purpose so that it can be reviewed against the rules above.
"""
import torch


def _broadcast(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def _out_size(src, index, dim, dim_size):
    size = list(src.size())
    if dim_size is not None:
        size[dim] = dim_size
    elif index.numel() == 0:
        size[dim] = 0
    else:
        size[dim] = int(index.max()) + 1
    return size


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    index = _broadcast(index, src, dim)
    if out is None:
        out = torch.zeros(_out_size(src, index, dim, dim_size), dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


def scatter_mul(src, index, dim=-1, out=None, dim_size=None):
    raise NotImplementedError("scatter_mul is not used on the hot path")


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_sum(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count = _broadcast(count, out, dim)
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode='floor')
    return out


class _ScatterMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim, dim_size):
        index = _broadcast(index, src, dim)
        size = _out_size(src, index, dim, dim_size)
        n_src = src.size(dim)
        if src.is_floating_point():
            lowest = torch.finfo(src.dtype).min
        else:
            lowest = torch.iinfo(src.dtype).min
        out = torch.full(size, lowest, dtype=src.dtype, device=src.device)
        out.scatter_reduce_(dim, index, src, 'amax', include_self=True)
        # first source position (along dim) that attains the maximum of its destination
        shape = [1] * src.dim()
        shape[dim] = n_src
        pos = torch.arange(n_src, device=src.device).view(shape).expand(src.size())
        hit = src == out.gather(dim, index)
        cand = torch.where(hit, pos, torch.full_like(pos, n_src))
        arg = torch.full(size, n_src, dtype=torch.long, device=src.device)
        arg.scatter_reduce_(dim, index, cand, 'amin', include_self=True)
        out = out.masked_fill(arg == n_src, 0)
        ctx.dim = dim
        ctx.n_src = n_src
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        (arg,) = ctx.saved_tensors
        size = list(grad_out.size())
        size[ctx.dim] = ctx.n_src + 1
        grad_in = torch.zeros(size, dtype=grad_out.dtype, device=grad_out.device)
        grad_in.scatter_(ctx.dim, arg, grad_out)
        return grad_in.narrow(ctx.dim, 0, ctx.n_src), None, None, None


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    assert out is None
    return _ScatterMax.apply(src, index, dim, dim_size)


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ('sum', 'add'):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == 'max':
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)
