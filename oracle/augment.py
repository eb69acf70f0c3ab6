"""ORACLE / TEST INFRASTRUCTURE ONLY: the reference's training-time geometry restated in numpy.

Reference: ``dataloader/transform.py:105-113`` = ExtCompose([ExtRandomScale((0.5, 2.0)), ExtRandomCrop(768,
pad_values=[ignore_idx, nseg], padding=(124,116,104), pad_if_needed=True), ExtRandomHorizontalFlip(), ExtToTensor,
ExtNormalize]) with the classes of ``dataloader/ext_transforms.py:172-192`` (scale), ``:443-520`` (pad + crop),
``:323-341`` (flip), ``:384-437`` (to tensor, normalise).  Those classes call torchvision's functional API on PIL images,
i.e. Pillow's ``Image.resize`` (BILINEAR for the picture, NEAREST for label / superpixel maps), ``ImageOps.expand``,
``Image.crop`` and ``Image.transpose``.  torchvision is not installed in the build container and Pillow is a third-party
dependency (not under /root/reference), so its published resampling algorithm is restated here --
``src/libImaging/Resample.c`` (precompute_coeffs, normalize_coeffs_8bpc, the 8-bit horizontal and vertical passes) and
``src/libImaging/Geometry.c`` (ImagingScaleAffine, nearest) of Pillow 12 -- and pinned against Pillow itself, which IS
installed here: ``tests/golden/g9_augment.npz`` is produced by ``oracle/gen_golden.py:gen_g9`` with real PIL calls.

Random draws follow the reference's order on Python's ``random`` module: ``uniform`` (scale), then ``randint`` twice
(crop row, crop column; skipped when the padded image already has the crop size), then ``random`` (flip)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """Resample.c:precompute_coeffs + normalize_coeffs_8bpc, BILINEAR, box = the whole axis.
    -> (bounds[out,2] = (first source index, tap count), kk[out,ksize] int32 fixed-point weights)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            for x in range(xmax):
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, bounds, kk, axis):
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear_u8(img, out_h, out_w):
    """``Image.resize((out_w, out_h), BILINEAR)`` of an RGB u8 array [H,W,3]: horizontal pass, u8, vertical pass."""
    cur = img
    if out_w != img.shape[1]:
        cur = _resample_axis(cur, *pil_bilinear_coeffs(img.shape[1], out_w), 1)
    if out_h != img.shape[0]:
        cur = _resample_axis(cur, *pil_bilinear_coeffs(img.shape[0], out_h), 0)
    return cur


def pil_nearest_index(in_size, out_size):
    """Source index per output index of ``Image.resize(..., NEAREST)`` (Geometry.c:ImagingScaleAffine: the source
    coordinate starts at a/2 and is advanced by repeated double-precision addition of a = in/out)."""
    a = in_size / out_size
    xo = a * 0.5
    idx = np.empty(out_size, dtype=np.int32)
    for x in range(out_size):
        idx[x] = min(int(xo), in_size - 1)
        xo += a
    return idx


def pil_resize_nearest(arr, out_h, out_w):
    return arr[pil_nearest_index(arr.shape[0], out_h)][:, pil_nearest_index(arr.shape[1], out_w)]


def draw_params(rng, H, W, crop, scale_range=(0.5, 2.0), p_flip=0.5):
    """The reference's random draws in its order; ``rng`` is a ``random.Random`` (or the ``random`` module)."""
    scale = rng.uniform(scale_range[0], scale_range[1])                 # ext_transforms.py:186
    th, tw = int(H * scale), int(W * scale)                             # :187
    gap_y = int(math.ceil((crop[0] - th) / 2)) if th < crop[0] else 0   # :488-490
    gap_x = int(math.ceil((crop[1] - tw) / 2)) if tw < crop[1] else 0   # :496-498
    ph, pw = th + 2 * gap_y, tw + 2 * gap_x
    if pw == crop[1] and ph == crop[0]:                                 # :470-471
        i = j = 0
    else:
        i = rng.randint(0, ph - crop[0])                                # :473
        j = rng.randint(0, pw - crop[1])                                # :474
    flip = rng.random() < p_flip                                        # :339
    return dict(scale=scale, th=th, tw=tw, gap_y=gap_y, gap_x=gap_x, i=i, j=j, flip=bool(flip))


def train_augment(img, maps, pad_values, p, crop, mean, std, fill=(124, 116, 104)):
    """img u8 [H,W,3]; maps: list of integer arrays [H,W]; p = draw_params(...).
    -> (float32 [3,ch,cw] normalised image, list of int64 maps [ch,cw])."""
    th, tw, gy, gx = p['th'], p['tw'], p['gap_y'], p['gap_x']
    im = pil_resize_bilinear_u8(img, th, tw)
    ms = [pil_resize_nearest(m, th, tw) for m in maps]
    if gy or gx:
        canvas = np.empty((th + 2 * gy, tw + 2 * gx, 3), dtype=np.uint8)
        canvas[:] = np.asarray(fill, dtype=np.uint8)
        canvas[gy:gy + th, gx:gx + tw] = im
        im = canvas
        padded = []
        for m, v in zip(ms, pad_values):
            c = np.full((th + 2 * gy, tw + 2 * gx), v, dtype=np.int64)
            c[gy:gy + th, gx:gx + tw] = m
            padded.append(c)
        ms = padded
    i, j = p['i'], p['j']
    im = im[i:i + crop[0], j:j + crop[1]]
    ms = [np.asarray(m[i:i + crop[0], j:j + crop[1]], dtype=np.int64) for m in ms]
    if p['flip']:
        im = im[:, ::-1]
        ms = [m[:, ::-1] for m in ms]
    t = im.transpose(2, 0, 1).astype(np.float32) / np.float32(255)      # to_tensor
    t = (t - np.asarray(mean, dtype=np.float32)[:, None, None]) / np.asarray(std, dtype=np.float32)[:, None, None]
    return np.ascontiguousarray(t, dtype=np.float32), [np.ascontiguousarray(m) for m in ms]
