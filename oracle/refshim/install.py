"""Make the reference's own Python importable in the BUILD CONTAINER (never on the GPU box).

ORACLE / TEST INFRASTRUCTURE ONLY -- used by ``oracle/gen_golden.py`` (which writes ``tests/golden/*.npz``); nothing under
``tests/`` imports the reference at run time.

The reference (``/root/reference``) imports third-party packages that are not installed here
(``torch_scatter``, ``wandb``, ``skimage``) and a data layer that cannot be imported offline
(``dataloader/region_cityscapes.py:13`` downloads a plugin at import time).  ``install()`` puts
small stand-ins into ``sys.modules`` *before* the reference modules are imported:

* ``torch_scatter``     -> ``oracle/refshim/torch_scatter`` (restated scatter semantics)
* ``dataloader``        -> empty package exposing ``get_dataset`` / ``get_slide_dataset`` names
  (read at import by ``trainer/base.py:7``)
* ``dataloader.utils``  -> ``collate_fn`` (same stacking rule as ``dataloader/utils.py:10-25``
  for the keys the scorer uses) and a ``DataProvider`` placeholder (``trainer/base.py:11``)
* ``wandb``             -> empty module
* ``skimage`` (+ ``.morphology.binary_dilation`` = scipy's, ``.segmentation.mark_boundaries``)
* ``cv2``               -> ``copyMakeBorder`` (constant border = ``numpy.pad``) and ``resize`` restricted to the
  identity case (same size in and out), the only one ``utils/sliding_evaluator*.py`` reaches on its working branch;
  ``collections.Iterable`` (removed in Python 3.10, used at ``sliding_evaluator.py:51``) is aliased to ``collections.abc``

No reference source is copied; the reference is imported from where it lies.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MULACTSEG_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "active_selection"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)                # -> import torch_scatter resolves to the stand-in
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)      # -> import utils.loss, trainer.*, active_selection.*

    import torch

    def collate_fn(inputs):
        out = {}
        for key in inputs[0].keys():
            vals = [b[key] for b in inputs]
            if isinstance(vals[0], torch.Tensor):
                out[key] = torch.stack(vals)
            elif type(vals[0]).__module__ == 'numpy':
                out[key] = torch.stack([torch.from_numpy(v) for v in vals])
            else:
                out[key] = vals
        return out

    class DataProvider:  # placeholder: never instantiated by the oracle
        pass

    pkg = _module("dataloader", get_dataset=None, get_slide_dataset=None, get_active_dataset=None)
    pkg.__path__ = []
    pkg.utils = _module("dataloader.utils", collate_fn=collate_fn, DataProvider=DataProvider)
    _module("wandb")

    try:
        from scipy import ndimage

        def binary_dilation(image, footprint=None, selem=None, out=None):
            st = footprint if footprint is not None else selem
            return ndimage.binary_dilation(image, structure=st)
    except Exception:  # pragma: no cover
        binary_dilation = None
    import collections
    import collections.abc
    import numpy as _np
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable

    def copyMakeBorder(img, top, bottom, left, right, border_mode, value=0):
        assert border_mode == 0
        pad = [(int(top), int(bottom)), (int(left), int(right))] + [(0, 0)] * (img.ndim - 2)
        return _np.pad(img, pad, mode="constant", constant_values=value)

    def resize(arr, dsize, interpolation=1):
        assert (arr.shape[1], arr.shape[0]) == tuple(dsize), "cv2 stand-in: only the identity resize is restated"
        return arr
    _module("cv2", copyMakeBorder=copyMakeBorder, resize=resize, BORDER_CONSTANT=0, INTER_LINEAR=1)

    sk = _module("skimage")
    sk.__path__ = []
    sk.morphology = _module("skimage.morphology", binary_dilation=binary_dilation)
    sk.segmentation = _module("skimage.segmentation", mark_boundaries=lambda im, *a, **k: im)
    return REFERENCE_ROOT
