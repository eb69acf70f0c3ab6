"""Make the reference's own Python importable in the BUILD CONTAINER (never on the GPU box).

ORACLE / TEST INFRASTRUCTURE ONLY -- used by ``oracle/gen_golden.py`` and by the optional
``tests/test_reference_live.py`` (skipped when ``/root/reference`` is absent).

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
    sk = _module("skimage")
    sk.__path__ = []
    sk.morphology = _module("skimage.morphology", binary_dilation=binary_dilation)
    sk.segmentation = _module("skimage.segmentation", mark_boundaries=lambda im, *a, **k: im)
    return REFERENCE_ROOT
