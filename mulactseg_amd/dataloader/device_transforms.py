"""Training-time augmentation on the device.

Mirror of the reference's training transform ``dataloader/transform.py:105-113``::

    ExtCompose([ExtRandomScale((0.5, 2.0)),
                ExtRandomCrop(size=(768, 768), pad_values=[ignore_idx, nseg], padding=(124, 116, 104), pad_if_needed=True),
                ExtRandomHorizontalFlip(), ExtToTensor(dtype_list=[...]), ExtNormalize(mean, std)])

(``dataloader/ext_transforms.py:172-192, 443-520, 323-341, 384-437``) for pictures and maps that are already resident
in HBM (decoded once; a Cityscapes picture is 6 MB as u8, the whole 2 975-image training set 18.7 GB).  The reference
runs these steps on PIL images in 12 DataLoader workers; here one kernel per sample (``csrc/augment.hip``) writes the
normalised float crop and the cropped maps directly.  Results are bit-identical to Pillow's ``Image.resize``
(BILINEAR / NEAREST) followed by pad, crop, flip, ``to_tensor`` and ``normalize``: the host computes Pillow's
fixed-point coefficient tables in double precision exactly as ``src/libImaging/Resample.c`` does, the kernel applies
them in integer arithmetic.  The random draws are the reference's, in its order, on Python's ``random``:
``uniform`` (scale), ``randint`` x 2 (crop origin; skipped when nothing is left to choose), ``random`` (flip)."""
import math
import random as _random

import numpy as np
import torch

from .. import _lib

PRECISION_BITS = 32 - 8 - 2
_MAP_CODES = {torch.int64: _lib.ID_I64, torch.int32: _lib.ID_I32, torch.int16: _lib.ID_U16, torch.uint8: 3}
if hasattr(torch, "uint16"):
    _MAP_CODES[torch.uint16] = _lib.ID_U16


def bilinear_tables(in_size, out_size):
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for BILINEAR over a whole axis, vectorised over the
    output index with the per-element operation order of the C loop.  -> (bounds int32 [out,2], kk int32 [out,ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # (int) truncates toward zero
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    w = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):
        live = x < xmax
        v = np.abs((x + xmin - center + 0.5) * ss)
        f = np.where(live & (v < 1.0), 1.0 - v, 0.0)
        w[:, x] = f
        ww = ww + f                                                          # sequential, as in the C loop
    nz = ww != 0.0
    w[nz] = w[nz] / ww[nz, None]
    kk = (0.5 + w * float(1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)   # weights are >= 0: truncation = floor
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk


def nearest_table(in_size, out_size):
    """Source index per output index of Pillow's NEAREST resize (``Geometry.c:ImagingScaleAffine``): the coordinate
    starts at a/2 and advances by repeated double additions of a = in/out (a cumulative sum, not a multiplication)."""
    a = in_size / out_size
    steps = np.full(out_size, a, dtype=np.float64)
    steps[0] = a * 0.5
    xo = np.cumsum(steps)
    return np.minimum(xo.astype(np.int64), in_size - 1).astype(np.int32)


class _PinnedRing:
    """Page-locked staging slots for the per-sample coefficient tables: a copy from pageable memory blocks the host until
    the stream's earlier work has drained, so a prefetching provider would serialise with its own augmentation kernels;
    from a pinned slot the copy is only enqueued.  A slot is reused after the event recorded behind its copy."""

    def __init__(self, slots=8):
        self.bufs, self.events, self.k = [None] * slots, [None] * slots, 0

    def upload(self, arr, device):
        k, self.k = self.k, (self.k + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        n = int(arr.size)
        if self.bufs[k] is None or self.bufs[k].numel() < n:
            self.bufs[k] = torch.empty(max(n, 1 << 15), dtype=torch.int32).pin_memory()
        host = self.bufs[k][:n]
        host.numpy()[:] = arr
        dev = torch.empty(n, dtype=torch.int32, device=device)
        dev.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[k] = ev
        return dev


def draw_params(rng, H, W, crop, scale_range=(0.5, 2.0), p_flip=0.5):
    """The reference's random draws, in its order (``ext_transforms.py:186-187, 470-474, 339``)."""
    scale = rng.uniform(scale_range[0], scale_range[1])
    th, tw = int(H * scale), int(W * scale)
    gap_y = int(math.ceil((crop[0] - th) / 2)) if th < crop[0] else 0
    gap_x = int(math.ceil((crop[1] - tw) / 2)) if tw < crop[1] else 0
    ph, pw = th + 2 * gap_y, tw + 2 * gap_x
    if pw == crop[1] and ph == crop[0]:
        i = j = 0
    else:
        i = rng.randint(0, ph - crop[0])
        j = rng.randint(0, pw - crop[1])
    flip = rng.random() < p_flip
    return dict(scale=scale, th=th, tw=tw, gap_y=gap_y, gap_x=gap_x, i=i, j=j, flip=bool(flip))


class DeviceTrainAugment:
    """``transform(img_u8[H,W,3] cuda, [map, ...] cuda) -> (image f32 [3,ch,cw], [maps])`` -- up to two maps
    (label, superpixel ids), int64 out (uint8 in -> uint8 out when ``keep_u8``)."""

    def __init__(self, size=(768, 768), scale_range=(0.5, 2.0), pad_values=(255, 2048), fill=(124, 116, 104),
                 mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), rng=None, keep_u8=True):
        self.size = (int(size), int(size)) if isinstance(size, int) else tuple(int(v) for v in size)
        self.scale_range = scale_range
        self.pad_values = list(pad_values)
        self.fill = (np.asarray(fill, dtype=np.uint8))
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)
        self.rng = rng if rng is not None else _random
        self.keep_u8 = keep_u8
        self._ring = _PinnedRing()

    def __call__(self, img, maps=(), params=None):
        if not (img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.is_contiguous()):
            raise ValueError("picture must be a contiguous uint8 [H,W,3] tensor on the GPU")
        maps = list(maps)
        if len(maps) > 2 or len(maps) > len(self.pad_values):
            raise ValueError("at most two maps, each with a pad value")
        H, W = int(img.shape[0]), int(img.shape[1])
        for m in maps:
            if not (m.is_cuda and tuple(m.shape) == (H, W) and m.dtype in _MAP_CODES and m.is_contiguous()):
                raise ValueError("maps must be contiguous [H,W] integer tensors on the picture's device")
        p = params if params is not None else draw_params(self.rng, H, W, self.size, self.scale_range)
        th, tw = p['th'], p['tw']
        hb, hk = bilinear_tables(W, tw)
        vb, vk = bilinear_tables(H, th)
        xi, yi = nearest_table(W, tw), nearest_table(H, th)
        parts = [hb.ravel(), hk.ravel(), vb.ravel(), vk.ravel(), xi, yi]
        offs = np.cumsum([0] + [a.size for a in parts])
        tab = self._ring.upload(np.concatenate(parts).astype(np.int32, copy=False), img.device)   # one small pinned H2D copy
        ptr = [tab.data_ptr() + 4 * int(o) for o in offs[:-1]]
        ch, cw = self.size
        out = torch.empty((3, ch, cw), dtype=torch.float32, device=img.device)
        outs, margs = [], []
        for k in range(2):
            if k < len(maps):
                u8 = self.keep_u8 and maps[k].dtype == torch.uint8
                o = torch.empty((ch, cw), dtype=torch.uint8 if u8 else torch.int64, device=img.device)
                outs.append(o)
                margs += [maps[k].data_ptr(), _MAP_CODES[maps[k].dtype], int(self.pad_values[k]), o.data_ptr(), int(u8)]
            else:
                margs += [None, 0, 0, None, 0]
        lib = _lib.load()
        with torch.cuda.device(img.device):
            st = torch.cuda.current_stream(img.device).cuda_stream
            _lib.check(lib.mas_train_augment(img.data_ptr(), H, W, th, tw, ptr[0], ptr[1], hk.shape[1], ptr[2], ptr[3], vk.shape[1],
                                             ptr[4], ptr[5], p['gap_y'], p['gap_x'], p['i'], p['j'], int(p['flip']), ch, cw,
                                             self.mean.ctypes.data, self.std.ctypes.data, self.fill.ctypes.data, *margs,
                                             out.data_ptr(), st), "mas_train_augment")
        tab.record_stream(torch.cuda.current_stream(img.device))
        return out, outs


def get_device_transform(args):
    """The reference's ``'rescale_769_multi'`` training transform (``transform.py:67-89``) for resident data."""
    return DeviceTrainAugment(size=(768, 768), scale_range=(0.5, 2.0), pad_values=[args.ignore_idx, args.nseg])


class DeviceResize(DeviceTrainAugment):
    """The reference's deterministic transforms for pool / validation / evaluation pictures on the same kernel:
    ``ExtResize((h, w))`` (``dataloader/__init__.py:124-136``: Cityscapes to 1024x2048 -- the identity for native pictures, every
    bilinear weight is then exactly 1) or ``ExtResize(s)`` + ``ExtCenterCrop(s)`` (VOC, ``:156-170``: the shorter side to ``s``,
    torchvision's ``int(s * long / short)`` for the other, crop origin ``int(round((side - s) / 2.))``), then to-tensor + normalise.
    No random draw is consumed."""

    def __init__(self, size, center_crop=None, pad_values=(255, 2048), **kw):
        super().__init__(size=(1, 1), scale_range=(1.0, 1.0), pad_values=pad_values, **kw)
        self.target = size
        self.center_crop = center_crop

    def geometry(self, H, W):
        if isinstance(self.target, int):
            s = int(self.target)
            if W <= H:
                tw, th = s, int(s * H / W)
            else:
                th, tw = s, int(s * W / H)
        else:
            th, tw = int(self.target[0]), int(self.target[1])
        if self.center_crop is None:
            return dict(scale=1.0, th=th, tw=tw, gap_y=0, gap_x=0, i=0, j=0, flip=False), (th, tw)
        c = int(self.center_crop)
        if th < c or tw < c:
            raise ValueError("center crop %d larger than the resized picture %dx%d" % (c, th, tw))
        return dict(scale=1.0, th=th, tw=tw, gap_y=0, gap_x=0, i=int(round((th - c) / 2.)), j=int(round((tw - c) / 2.)), flip=False), (c, c)

    def __call__(self, img, maps=(), params=None):
        p, self.size = self.geometry(int(img.shape[0]), int(img.shape[1]))
        return super().__call__(img, maps, params=p)


class DeviceResizeFlip(DeviceResize):
    """``ExtResize(s)`` + ``ExtCenterCrop(s)`` + ``ExtRandomHorizontalFlip`` -- the VOC stage-2 training transform
    (``transform_voc.py:52-61``, name ``rescale_769_nospx``): one ``random()`` draw per sample."""

    def __call__(self, img, maps=(), params=None):
        p, self.size = self.geometry(int(img.shape[0]), int(img.shape[1]))
        p['flip'] = bool(self.rng.random() < 0.5)
        return DeviceTrainAugment.__call__(self, img, maps, params=p)
