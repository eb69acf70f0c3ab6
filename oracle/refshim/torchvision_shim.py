"""ORACLE / TEST INFRASTRUCTURE ONLY: the handful of ``torchvision.transforms.functional`` entry points that the reference's
``dataloader/ext_transforms.py`` calls on PIL images, restated as the Pillow calls torchvision 0.12 makes for them
(``torchvision/transforms/functional_pil.py``; torchvision is not installed in the build container, Pillow -- the code that does
the arithmetic -- is).  Installed into ``sys.modules`` by ``install_torchvision()`` so that the reference's own transform classes
(ExtRandomScale, ExtRandomCrop, ExtRandomHorizontalFlip, ExtToTensor, ExtNormalize) can be imported and RUN to produce G9:
the order of the random draws then comes from the reference, not from a restatement.

Parity note: this file is a stand-in at a third-party boundary (torchvision==0.12.0, ``actsegmul.yml``); what it pins is the
reference's use of it, not torchvision itself.
"""
import enum
import sys
import types

import numpy as np
import torch
from PIL import Image, ImageOps


class InterpolationMode(enum.Enum):
    NEAREST = "nearest"
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"


_PIL = {InterpolationMode.NEAREST: Image.NEAREST, InterpolationMode.BILINEAR: Image.BILINEAR, InterpolationMode.BICUBIC: Image.BICUBIC}


def _pil_mode(interpolation):
    return _PIL[interpolation] if isinstance(interpolation, InterpolationMode) else interpolation      # (an int is passed through)


def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
    """functional_pil.resize for a (h, w) sequence: ``img.resize(size[::-1], interpolation)``."""
    if not isinstance(size, (list, tuple)) or len(size) != 2:
        raise NotImplementedError("the reference passes (h, w)")
    return img.resize((int(size[1]), int(size[0])), _pil_mode(interpolation))


def pad(img, padding, fill=0, padding_mode="constant"):
    """functional_pil.pad, constant mode: ``ImageOps.expand(img, border=(left, top, right, bottom), fill=fill)``."""
    if padding_mode != "constant":
        raise NotImplementedError
    if isinstance(padding, int):
        padding = (padding,) * 4
    elif len(padding) == 2:
        padding = (padding[0], padding[1], padding[0], padding[1])
    if isinstance(fill, (list, tuple)) and len(img.getbands()) == 1:
        fill = fill[0]
    if isinstance(fill, (int, float)) and len(img.getbands()) > 1:
        fill = tuple([int(fill)] * len(img.getbands()))
    return ImageOps.expand(img, border=tuple(int(p) for p in padding), fill=tuple(fill) if isinstance(fill, (list, tuple)) else fill)


def crop(img, top, left, height, width):
    return img.crop((left, top, left + width, top + height))


def center_crop(img, output_size):
    th, tw = (output_size, output_size) if isinstance(output_size, int) else output_size
    w, h = img.size
    return crop(img, int(round((h - th) / 2.0)), int(round((w - tw) / 2.0)), th, tw)


def hflip(img):
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def vflip(img):
    return img.transpose(Image.FLIP_TOP_BOTTOM)


def to_tensor(pic):
    """functional.to_tensor for a uint8 PIL image: HWC -> CHW, float32, / 255."""
    arr = np.array(pic, copy=True)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(arr).permute(2, 0, 1).contiguous()
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


def normalize(tensor, mean, std, inplace=False):
    t = tensor if inplace else tensor.clone()
    m = torch.as_tensor(mean, dtype=t.dtype)[:, None, None]
    s = torch.as_tensor(std, dtype=t.dtype)[:, None, None]
    return t.sub_(m).div_(s)


def rotate(*a, **k):
    raise NotImplementedError("ExtRandomRotation is not on the path")


def install_torchvision():
    """Put the stand-in under the names ``dataloader/ext_transforms.py:3-12`` imports."""
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")
    for name in ("resize", "pad", "crop", "center_crop", "hflip", "vflip", "to_tensor", "normalize", "rotate"):
        setattr(fn, name, globals()[name])
    fn.InterpolationMode = InterpolationMode
    tr.InterpolationMode = InterpolationMode
    tr.functional = fn
    tv.transforms = tr
    tv.__path__ = []
    tr.__path__ = []
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr
    sys.modules["torchvision.transforms.functional"] = fn
    return tv
