"""The stream-K hand-off of csrc/conv_sk.hip beside OTHER resident kernels -- what DistributedDataParallel's RCCL all-reduce does to
the input-gradient kernels during backward (reference trainer/base.py:27-35 wraps the net in DataParallel).

The design (include/mulactseg_hip.h, "co-residency contract"): contributors publish first and never wait, finishers wait at the end of
their work for higher-indexed contributors only, so a launch completes whenever its workgroups are eventually scheduled -- it does not
need all 256 of them resident at once.  Checked here:
  * integer-exact products stay bit-exact while a second stream holds 32 / 96 / 224 CUs (96 KB of LDS per occupier: no stream-K
    workgroup fits beside one) for the whole duration of the launches, error word 0;
  * a give-up -- provoked with a one-poll wait bound under the heaviest occupation -- sets the error word AND poisons the tile with NaN
    (never plausible numbers), and the whole-tile plan (MAS_SK_NOSPLIT) reproduces the reference on the re-run;
  * one training step under torch.distributed (nccl = RCCL, world size 1) + DistributedDataParallel: gradients bit-identical to the
    step without it, error word 0;
  * the trainers raise StreamKGaveUp when the word is set (read with the loss, no extra synchronisation)."""
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import occupy_cus

pytestmark = pytest.mark.gpu

SPLIT_CASES = [
    # Cin, Cout, k, stride, dil, N, H, W: every one has tiles shared by two or three workgroups on 256 CUs
    (1024, 256, 1, 1, 1, 4, 48, 48), (256, 256, 3, 1, 1, 4, 48, 48), (512, 512, 3, 1, 2, 2, 48, 48), (64, 64, 3, 1, 1, 1, 96, 96),
    (304, 256, 1, 1, 1, 1, 64, 64), (256, 256, 3, 1, 1, 2, 49, 49), (512, 128, 1, 1, 1, 2, 97, 97), (128, 128, 3, 2, 1, 2, 64, 96),
]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _integer_case(g, Cin, Cout, k, stride, dil, N, H, W):
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g, device='cuda').float()
    w = torch.randint(-2, 3, (Cout, Cin, k, k), generator=g, device='cuda').float()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randint(-2, 3, (N, Cout, Ho, Wo), generator=g, device='cuda').float()
    xd, wd = x.double().requires_grad_(True), w.double()
    y = F.conv2d(xd, wd, None, stride, dil if k == 3 else 0, dil)
    y.backward(dy.double())
    return x, w, dy, y.detach(), xd.grad


@pytest.mark.parametrize("occupied", [32, 96, 224])
def test_exact_beside_a_cu_hogging_stream(occupied):
    _need_gpu()
    from mulactseg_amd import ops
    g = torch.Generator(device='cuda').manual_seed(17 + occupied)
    cases = [_integer_case(g, *c) for c in SPLIT_CASES]
    ops.conv_sk_clear_error()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    # one occupier per CU (96 KB of LDS: neither a second occupier nor a stream-K workgroup fits beside it), alive for 0.4 s -- the
    # launches below are all queued (and most of them run) inside that window; `busy` checks that at least the first ones did
    occupy_cus(occupied, 96 * 1024, 0.4, stream=side)
    done_side = torch.cuda.Event()
    outs = []
    for (c, (x, w, dy, y_ref, dx_ref)) in zip(SPLIT_CASES, cases):
        stride, dil = c[3], c[4]
        for rep in range(2):
            outs.append((c, "fwd", ops.conv_sk(x, w, stride, dil), y_ref))
            if stride == 1:
                outs.append((c, "dgrad", ops.conv_sk(dy, w, 1, dil, dgrad=True), dx_ref))
    first_done = torch.cuda.Event()
    first_done.record()
    with torch.cuda.stream(side):
        done_side.record()
    first_done.synchronize()
    busy = not done_side.query()            # the occupiers were still there when the last product finished
    torch.cuda.synchronize()
    for c, role, got, ref in outs:
        assert torch.equal(got.double(), ref), (occupied, c, role)
    assert ops.conv_sk_error() == 0
    print("occupied %d CUs: %d products exact; occupiers outlived the products: %s" % (occupied, len(outs), busy))


def test_give_up_is_loud_and_the_whole_tile_plan_recovers():
    """A finisher whose wait bound runs out must not hand out numbers: error word + NaN tile.  Provoked with a bound of ONE poll while
    224 CUs are held (most contributors are then not even scheduled when their finisher looks); MAS_SK_NOSPLIT -- no hand-off, no
    wait -- reproduces the reference."""
    _need_gpu()
    from mulactseg_amd import _lib, ops
    g = torch.Generator(device='cuda').manual_seed(23)
    side = torch.cuda.Stream()
    provoked = 0
    for c in SPLIT_CASES:
        x, w, dy, y_ref, dx_ref = _integer_case(g, *c)
        stride, dil = c[3], c[4]
        ops.conv_sk_clear_error()
        torch.cuda.synchronize()
        occupy_cus(224, 96 * 1024, 0.05, stream=side)
        y = ops.conv_sk(x, w, stride, dil, spin_limit=1)
        err = ops.conv_sk_error()                       # (synchronises)
        if err:
            provoked += 1
            assert bool(torch.isnan(y).any()), "a launch that gave up must poison its tile"
            bad = torch.isnan(y)
            assert torch.equal(y[~bad].double(), y_ref[~bad]), "tiles that did not give up are exact"
            words = ops.conv_sk_error_words()
            assert words is not None and int(words.max()) != 0
            ops.conv_sk_clear_error()
            assert ops.conv_sk_error() == 0
        else:
            assert torch.equal(y.double(), y_ref)
        torch.cuda.synchronize()
        # the whole-tile plan, with and without the neighbour: no hand-off at all, so the one-poll bound cannot matter
        occupy_cus(224, 96 * 1024, 0.05, stream=side)
        y2 = ops.conv_sk(x, w, stride, dil, flags=_lib.SK_NOSPLIT, spin_limit=1)
        assert torch.equal(y2.double(), y_ref), c
        if stride == 1:
            dx2 = ops.conv_sk(dy, w, 1, dil, dgrad=True, flags=_lib.SK_NOSPLIT, spin_limit=1)
            assert torch.equal(dx2.double(), dx_ref), c
        assert ops.conv_sk_error() == 0
    torch.cuda.synchronize()
    print("give-up provoked in %d of %d launches" % (provoked, len(SPLIT_CASES)))


def test_whole_tile_plan_matches_on_random_data():
    _need_gpu()
    from mulactseg_amd import _lib, ops
    torch.manual_seed(3)
    for Cin, Cout, k, stride, dil, N, H, W in SPLIT_CASES + [(64, 200, 3, 1, 1, 1, 9, 33), (256, 48, 1, 1, 1, 2, 20, 36), (64, 64, 3, 1, 1, 1, 385, 385)]:
        x = torch.randn(N, Cin, H, W, device='cuda')
        w = torch.randn(Cout, Cin, k, k, device='cuda')
        a = ops.conv_sk(x, w, stride, dil)
        b = ops.conv_sk(x, w, stride, dil, flags=_lib.SK_NOSPLIT)
        # (a split tile adds its parts in chunk order: the same k order, but partial sums are rounded where the parts meet)
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
        ya, pa = ops.conv_sk(x, w, stride, dil, stats=True)
        yb, pb = ops.conv_sk(x, w, stride, dil, stats=True, flags=_lib.SK_NOSPLIT)
        assert torch.equal(yb, b) and torch.equal(ya, a)
        # (partials are f32 sums of the tile's accumulators: the two plans round at different places)
        assert float((pa.sum(1) - pb.sum(1)).abs().max()) <= 1e-5 * float(pa.sum(1)[:, 1].max())
    assert ops.conv_sk_error() == 0


def _one_step(net, x, wts):
    for p in net.parameters():
        p.grad = None
    z = net(x, lowres=True)
    (z * wts).sum().backward()
    return z


def test_training_step_under_rccl_ddp_is_bit_identical():
    """DistributedDataParallel (backend nccl = RCCL) launches its bucketed all-reduce kernels on its own stream WHILE the backward
    pass runs the stream-K input-gradient kernels.  World size 1 on this one-GPU box: the same hooks, buckets, streams and RCCL
    kernels as with N ranks.  Gradients equal the plain step bit for bit, error word 0."""
    _need_gpu()
    import torch.distributed as dist
    from mulactseg_amd import ops
    from mulactseg_amd.models import get_model
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dev = torch.device('cuda:0')
    torch.manual_seed(13)
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    bn_state = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator(device=dev).manual_seed(4), device=dev)
    ops.conv_sk_clear_error()
    wts = torch.linspace(-1.0, 1.0, 2 * 20 * 64 * 64, device=dev).view(2, 20, 64, 64)      # quarter-resolution logits
    _one_step(net, x, wts)                                      # (first call of the process: MIOpen's find may time other solvers)
    net.load_state_dict(bn_state)
    z_plain = _one_step(net, x, wts).detach().clone()
    g_plain = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    net.load_state_dict(bn_state)
    z_again = _one_step(net, x, wts).detach()
    assert torch.equal(z_again, z_plain), ("the plain step is not run-to-run identical", float((z_again - z_plain).abs().max()))
    net.load_state_dict(bn_state)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29613")
    import datetime
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], bucket_cap_mb=16, gradient_as_bucket_view=True)
        for rep in range(2):
            net.load_state_dict(bn_state)
            z_ddp = _one_step(ddp, x, wts).detach()
            torch.cuda.synchronize()
            assert torch.equal(z_ddp, z_plain), float((z_ddp - z_plain).abs().max())
            for n, p in net.named_parameters():
                if n in g_plain:
                    assert torch.equal(p.grad, g_plain[n]), (rep, n)
        assert ops.conv_sk_error() == 0
    finally:
        dist.destroy_process_group()


def test_trainer_raises_when_the_error_word_is_set():
    """trainer/base.py: stream_k_flag() rides on the loss's host read; a set word raises StreamKGaveUp and is cleared."""
    _need_gpu()
    from mulactseg_amd import ops
    from mulactseg_amd.trainer.base import BaseTrainer
    x = torch.randn(1, 64, 16, 16, device='cuda')
    w = torch.randn(64, 64, 1, 1, device='cuda')
    ops.conv_sk(x, w)                                # (the device's workspace exists)
    ops.conv_sk_clear_error()
    t = BaseTrainer.__new__(BaseTrainer)
    t.device = torch.device('cuda', torch.cuda.current_device())
    flag = t.stream_k_flag()
    assert len(flag) == 1 and int(flag[0]) == 0
    t.check_stream_k()
    for v in ops._sk_error_views(t.device):
        v.view(torch.int32).fill_(1)                 # what a finisher's atomicOr does
    assert int(t.stream_k_flag()[0]) == 1
    with pytest.raises(ops.StreamKGaveUp):
        t.check_stream_k()
    assert ops.conv_sk_error() == 0


def test_a_give_up_skips_the_optimizer_update_on_the_device():
    """trainer/base.py:guard_optimizer_step -- the flag of THIS step's passes reaches the fused AdamW kernel as `found_inf` (no host
    round trip): parameters, moments and step counts stay as they were; with the word clear the same step updates them."""
    _need_gpu()
    from mulactseg_amd import ops
    from mulactseg_amd.trainer.base import BaseTrainer
    dev = torch.device('cuda', torch.cuda.current_device())
    ops.conv_sk(torch.randn(1, 64, 16, 16, device=dev), torch.randn(64, 64, 1, 1, device=dev))      # (the workspace exists)
    ops.conv_sk_clear_error()
    p = torch.nn.Parameter(torch.randn(1000, device=dev))
    t = BaseTrainer.__new__(BaseTrainer)
    t.device = dev
    t.optimizer = torch.optim.AdamW([p], lr=0.1, fused=True)
    p.grad = torch.ones_like(p)
    t.guard_optimizer_step()
    t.optimizer.step()                               # word clear: a normal update
    after_one = p.detach().clone()
    assert float(t.optimizer.state[p]['step']) == 1.0
    for v in ops._sk_error_views(dev):
        v.view(torch.int32).fill_(1)                 # what a finisher's atomicOr does
    p.grad = torch.full_like(p, float('nan'))        # the gradients of a poisoned step
    t.guard_optimizer_step()
    t.optimizer.step()
    assert torch.equal(p.detach(), after_one) and float(t.optimizer.state[p]['step']) == 1.0
    assert bool(torch.isfinite(t.optimizer.state[p]['exp_avg']).all())
    ops.conv_sk_clear_error()
    p.grad = torch.ones_like(p)
    t.guard_optimizer_step()
    t.optimizer.step()
    assert not torch.equal(p.detach(), after_one) and float(t.optimizer.state[p]['step']) == 2.0
