"""mIoU counters on the GPU vs the golden produced by the reference's MeanIoU / IoUIgnore (g5) --
exact integer counters, identical IoU doubles."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g5():
    g = np.load(os.path.join(GOLDEN, "g5_miou.npz"))
    rs = np.random.RandomState(int(g['seed']))
    B, H, W, nc = int(g['B']), int(g['H']), int(g['W']), int(g['nc'])
    logits = rs.standard_normal(size=(2, B, nc + 1, H, W)).astype(np.float32)
    labels = rs.randint(0, nc, size=(2, B, H, W)).astype(np.int64)
    labels[labels == 7] = 3
    labels[rs.uniform(size=labels.shape) < 0.15] = 255
    return g, logits, labels, nc


def test_meters_match_reference_golden():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.utils.miou import IoUIgnore, LogitsIoU, MeanIoU
    g, logits, labels, nc = _g5()
    m, ig, fused = MeanIoU(nc, 255), IoUIgnore(num_classes=nc, ignore_label=255), LogitsIoU(nc, 255)
    m._before_epoch()
    for step in range(2):
        p = torch.from_numpy(logits[step]).cuda()
        t = torch.from_numpy(labels[step]).cuda()
        m._after_step({'outputs': p[:, :-1].max(dim=1)[1], 'targets': t})
        ig._after_step({'outputs': p.max(dim=1)[1], 'targets': t})
        fused.step(p, t)
    for meter in (m, fused):
        assert np.array_equal(meter.total_seen, g['seen'])
        assert np.array_equal(meter.total_correct, g['correct'])
        assert np.array_equal(meter.total_positive, g['positive'])
        ious = meter._after_epoch()
        assert np.array_equal(np.array(ious, dtype=np.float64), g['ious'])
        assert ious[7] == 100 and np.mean(ious) == g['miou']
    assert np.array_equal(np.array([ig.total_seen, ig.total_correct, ig.total_positive]), g['ign'])
    assert ig._after_epoch() == g['ign_iou'] and fused.ignore_iou() == g['ign_iou']


def test_fused_counts_equal_label_map_counts_at_full_resolution():
    """Size-independent property at the evaluation shape [2,20,1024,2048]: fused == two-step, and
    seen == number of non-ignored in-range pixels."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd import ops
    B, CH, H, W, nc = 2, 20, 1024, 2048, 19
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    z = torch.randn((B, CH, H, W), generator=g, device='cuda')
    t = torch.randint(0, nc, (B, H, W), generator=g, device='cuda')
    t[torch.rand((B, H, W), generator=g, device='cuda') < 0.1] = 255
    fused = ops.logits_iou_counts(z, t, nc, 255)
    two = ops.iou_counts(z[:, :-1].max(dim=1)[1], z.max(dim=1)[1], t, nc, 255)
    assert torch.equal(fused, two)
    assert int(fused[:nc].sum()) == int((t != 255).sum())
    assert int(fused[3 * nc]) == int((t == 255).sum())
