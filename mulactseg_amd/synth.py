"""Seeded synthetic Cityscapes/VOC-shaped inputs (no dataset exists on the GPU box).

Pure numpy (legacy ``RandomState`` streams are stable across numpy versions), so the golden
generator in the build container, the tests and ``bench.py`` on the GPU box all see the same bytes.
Shapes and value conventions follow the reference's tensor contract:

* logits ``z[B,C,H,W]`` f32 -- cosine-classifier range (about [-1.6, 1.6]) with a piecewise-constant
  class map so that top-2 margins are realistic under ``T = 0.1``
  (reference head: ``models/segmentation/deeplabv3.py:121-124``).
* superpixel id map ``spx[H,W]`` -- jittered grid of ``S`` compact cells, ids ``0..S-1``
  (pool images: every pixel valid, ``dataloader/__init__.py:121-126``); the training variant pads
  with the out-of-range id ``S`` (``dataloader/transform.py:107``).
* multi-hot targets ``[S,C]`` u8 with at least one bit per superpixel
  (precondition of ``trainer/active_joint_multi_predignore_lossdecomp.py:67``).
"""
import numpy as np


def grid_shape(S, H, W):
    """Pick gh x gw = S cells with roughly square cells for an H x W image."""
    best = None
    for gh in range(1, S + 1):
        if S % gh:
            continue
        gw = S // gh
        cost = abs((H / gh) - (W / gw))
        if best is None or cost < best[0]:
            best = (cost, gh, gw)
    return best[1], best[2]


def superpixel_map(seed, H, W, S, n_missing=0):
    """Jittered-grid superpixel ids [H,W] int64 in 0..S-1 (SEEDS-like compact cells)."""
    rs = np.random.RandomState(seed)
    gh, gw = grid_shape(S, H, W)
    ch, cw = H / gh, W / gw
    y = np.arange(H, dtype=np.float64)[:, None]
    x = np.arange(W, dtype=np.float64)[None, :]
    ph = rs.uniform(0, 2 * np.pi, size=4)
    ay, ax = 0.18 * ch, 0.18 * cw
    yy = y + ay * np.sin(2 * np.pi * x / (2.7 * cw) + ph[0]) + 0.5 * ay * np.sin(2 * np.pi * y / (1.9 * ch) + ph[1])
    xx = x + ax * np.sin(2 * np.pi * y / (3.1 * ch) + ph[2]) + 0.5 * ax * np.sin(2 * np.pi * x / (2.3 * cw) + ph[3])
    cy = np.clip(np.floor(yy / ch), 0, gh - 1).astype(np.int64)
    cx = np.clip(np.floor(xx / cw), 0, gw - 1).astype(np.int64)
    ids = cy * gw + cx
    if n_missing:
        # a few ids absent from the image (11 Cityscapes images lack one id): merge into a neighbour
        gone = rs.choice(S, size=n_missing, replace=False)
        for g in gone:
            ids[ids == g] = (g + 1) % S
    return ids


def class_map(seed, H, W, C, blob=24):
    """Piecewise-constant class layout [H,W] in 0..C-1 (coarse random grid, nearest upsample)."""
    rs = np.random.RandomState(seed)
    gh, gw = max(1, H // blob), max(1, W // blob)
    coarse = rs.randint(0, C, size=(gh + 1, gw + 1))
    yi = np.minimum((np.arange(H) * (gh + 1)) // H, gh)
    xi = np.minimum((np.arange(W) * (gw + 1)) // W, gw)
    return coarse[yi[:, None], xi[None, :]]


def logits(seed, B, C, H, W, noise=0.35, boost=0.6):
    """Cosine-like logits [B,C,H,W] f32: noise*N(0,1) + boost on the pixel's class."""
    rs = np.random.RandomState(seed)
    z = (noise * rs.standard_normal(size=(B, C, H, W))).astype(np.float32)
    for b in range(B):
        cm = class_map(seed * 7919 + b + 1, H, W, C)
        bb = np.zeros((C, H, W), dtype=np.float32)
        np.put_along_axis(bb, cm[None], np.float32(boost), axis=0)
        z[b] += bb
    return z


def multi_hot_targets(seed, S, C, p_counts=(0.70, 0.22, 0.06, 0.02)):
    """[S,C] u8 multi-hot rows, #bits drawn from {1,2,3,4} with the given probabilities."""
    rs = np.random.RandomState(seed)
    t = np.zeros((S, C), dtype=np.uint8)
    k = rs.choice(len(p_counts), size=S, p=np.asarray(p_counts) / np.sum(p_counts)) + 1
    for s in range(S):
        t[s, rs.choice(C, size=min(int(k[s]), C), replace=False)] = 1
    return t


def train_crop(seed, H, W, S, frac_selected=0.09, pad_frac=0.12):
    """Training-style (spx[H,W] int64 with pad id S, spmask[H,W] bool) pair.

    Mirrors the reference's geometry contract: out-of-image pixels carry the id ``S`` and are never
    selected (``dataloader/transform.py:107``, ``region_cityscapes_or_tensor.py:88-89``).
    """
    rs = np.random.RandomState(seed)
    ids = superpixel_map(seed + 1, H, W, S)
    ph, pw = int(H * pad_frac * rs.uniform()), int(W * pad_frac * rs.uniform())
    if ph:
        ids[H - ph:, :] = S
    if pw:
        ids[:, W - pw:] = S
    selected = rs.uniform(size=S + 1) < frac_selected
    selected[S] = False
    return ids, selected[ids]


def synthetic_state_dict(shapes, seed=0):
    """Deterministic weights from (key name, shape): lets two implementations of the same architecture be
    loaded with identical parameters without storing 107 MB.  ``shapes``: dict key -> tuple.
    Conv / linear weights ~ N(0, 2/fan_in), BN weight in [0.8, 1.2], bias / running_mean small,
    running_var in [0.6, 1.4], counters 0."""
    import zlib
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7fffffff)
        if key.endswith('num_batches_tracked'):
            out[key] = np.zeros(shape, dtype=np.int64)
        elif key.endswith('running_var'):
            out[key] = rs.uniform(0.6, 1.4, size=shape).astype(np.float32)
        elif key.endswith('running_mean') or key.endswith('.bias'):
            out[key] = (0.05 * rs.standard_normal(size=shape)).astype(np.float32)
        elif len(shape) == 1:
            out[key] = rs.uniform(0.8, 1.2, size=shape).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[key] = (np.sqrt(2.0 / fan_in) * rs.standard_normal(size=shape)).astype(np.float32)
    return out


def tiny_window_net(seed, num_channels, feat_dim=256):
    """A two-convolution stand-in for the segmentation net (``net(x)`` -> scores, ``net.feat_forward(x)`` -> (features,
    scores)) whose 3x3 zero-padded convolutions make every output depend on where the window border lies: used to pin
    the sliding-window evaluators (tests/golden/g8) without the cost of a DeepLab forward per window."""
    import torch

    class _Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            rs = np.random.RandomState(seed)
            self.wf = torch.nn.Parameter(torch.from_numpy(rs.standard_normal((feat_dim, 3, 3, 3)).astype(np.float32) * 0.2), False)
            self.ws = torch.nn.Parameter(torch.from_numpy(rs.standard_normal((num_channels, 3, 3, 3)).astype(np.float32) * 0.2), False)

        def feat_forward(self, x):
            return torch.nn.functional.conv2d(x, self.wf, padding=1), torch.nn.functional.conv2d(x, self.ws, padding=1)

        def forward(self, x):
            return torch.nn.functional.conv2d(x, self.ws, padding=1)

    return _Net().eval()
