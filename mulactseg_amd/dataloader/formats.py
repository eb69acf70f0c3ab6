"""On-disk / wire formats of the reference's data layer, read into the tensors and
lists the hot-path plugins consume.  File readers only -- no PIL augmentation pipeline.

* datalist ``*.txt``: one image per line, three tab-separated paths relative to the data root
  (image, label, superpixel file) -- ``dataloader/region_cityscapes.py:51-76``.
* region dict ``*.dict`` (JSON): ``spx path -> [n_ids, [missing ids]]`` or ``spx path -> [ids...]``
  -- ``region_cityscapes.py:137-153``.
* superpixel files: ``.pkl``/``.npy`` pickled dict with key ``'labels'`` (int map) or an image file
  -- ``region_cityscapes.py:94-101``.
* ``multi_hot_cls.npy`` u8 ``[N_img, nseg, num_classes + 1]`` (+ ``sp_size.npy``), rows indexed through the label
  file stem of the target datalist -- ``region_cityscapes_or_tensor.py:27-46``; built by
  ``dataloader/region_cityscapes_tensor.py:38-86`` / ``tools/label_assignment_tensor.py``.
"""
import json
import os

import numpy as np
import torch


def load_region_dict(path):
    """spx path -> list of valid superpixel ids."""
    with open(path, 'r') as f:
        data = json.load(f)
    if not data:
        return {}
    first = next(iter(data.values()))
    if isinstance(first, int):
        raise ValueError("region dict maps to a bare count: superpixel ids are not continuous (reference asserts here)")
    if isinstance(first[1], list):          # [size, [missing ids]]
        out = {}
        for key, (size, missing) in data.items():
            gone = set(missing)
            out[key] = [i for i in range(size) if i not in gone]
        return out
    if isinstance(first[1], int):           # explicit id list
        return data
    raise NotImplementedError("unknown region dict layout")


def read_datalist(datalist, root, region_dict, known_ignore=False, prob_dominant=False):
    """(im_idx, suppix): ``im_idx`` = list of [image, label, superpixel] absolute paths in file order, ``suppix`` =
    superpixel path -> ids still in this split -- ``region_cityscapes.py:51-76``."""
    ids = load_region_dict(region_dict) if isinstance(region_dict, str) else region_dict
    root = os.path.expanduser(root)
    with open(datalist, 'r') as f:
        lines = f.read().splitlines()
    im_idx, suppix = [], {}
    for line in lines:
        if not line:
            continue
        if not known_ignore:
            line = line.replace('gtFine_dominant', 'gtFine_dominant_ignore')
        if prob_dominant:
            line = line.replace('gtFine_dominant', 'gtFine_dominant_ignore_sample')
        img, lbl, spx = line.split('\t')
        full = [os.path.join(root, img), os.path.join(root, lbl), os.path.join(root, spx)]
        im_idx.append(full)
        suppix[full[2]] = ids[spx]
    return im_idx, suppix


def id_to_index(trg_datalist):
    """label-file stem -> row of ``multi_hot_cls`` (``region_cityscapes_or_tensor.py:41-46``)."""
    with open(trg_datalist, 'r') as f:
        lines = [l for l in f.read().splitlines() if l]
    return {l.split('\t')[1].split('/')[-1].split('.')[0]: i for i, l in enumerate(lines)}


def open_spx(path):
    """Superpixel id map as an int64 array [H,W] (``region_cityscapes.py:94-101``)."""
    ext = path.split('.')[-1]
    if ext in ('png', 'jpg'):
        from PIL import Image
        return np.array(Image.open(path)).astype(np.int64)
    data = np.load(path, allow_pickle=True)
    return np.asarray(data['labels']).astype(np.int64)


def multi_hot_paths(root, spx_method, nseg, trim_kernel_size=None):
    name = "gtFine_multi_tensor" if trim_kernel_size is None else "gtFine_multi_tensor_trim_{0}x{0}".format(trim_kernel_size)
    base = '{}/superpixel_seed/cityscapes/{}_{}/train/{}'.format(root, spx_method, nseg, name)
    return base + '/multi_hot_cls.npy', base + '/sp_size.npy'


def multi_hot_from_labels(target, superpixel, ids, nseg, num_classes, ignore=255):
    """The offline label assignment of one image: ``(superpixel_cls u8 [nseg, num_classes+1], sizes int32 [nseg])``;
    class c is set when any pixel of the superpixel carries train id c, the last column when any pixel is ``ignore``;
    sizes are -1 for ids not in ``ids`` (``region_cityscapes_tensor.py:38-86`` without boundary trimming)."""
    target = np.asarray(target).reshape(-1)
    superpixel = np.asarray(superpixel).reshape(-1)
    cls = np.zeros((nseg, num_classes + 1), dtype=np.uint8)
    size = np.full((nseg,), -1, dtype=np.int32)
    keep = np.isin(superpixel, np.asarray(ids))
    sp, tg = superpixel[keep], target[keep]
    col = np.where(tg == ignore, num_classes, tg)
    cls[sp, col] = 1
    cnt = np.bincount(sp, minlength=nseg)[:nseg]
    present = np.zeros(nseg, dtype=bool)
    present[np.asarray(ids, dtype=np.int64)] = True
    size[present] = cnt[present]
    return cls, size


def selection_lut(selected_ids, nseg, device):
    """bool [nseg + 1]: entry i true when id i is selected; the last entry (the crop pad id ``nseg``) is never set.
    Built on the host and moved with one copy."""
    lut = np.zeros(nseg + 1, dtype=np.bool_)
    if len(selected_ids):
        lut[np.asarray(list(selected_ids), dtype=np.int64)] = True
    lut[nseg] = False
    return torch.from_numpy(lut).to(device)


def selection_mask(superpixel, selected_ids, nseg):
    """``np.isin(superpixel, selected ids)`` (``region_cityscapes_or_tensor.py:88-89``) as a table lookup that works on
    device tensors: ids outside [0, nseg) (the crop pad id ``nseg``) are never selected."""
    lut = selection_lut(selected_ids, nseg, superpixel.device)
    return lut[superpixel.clamp(min=0, max=nseg).long()] & (superpixel >= 0)
