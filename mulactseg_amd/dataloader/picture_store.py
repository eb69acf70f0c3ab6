"""Decoded pictures and id maps of a file-backed dataset, kept in HBM.

The reference decodes a PNG/JPEG and a superpixel file inside every ``__getitem__`` (``dataloader/region_cityscapes.py:103-108``) and
hides the cost behind 12 DataLoader worker processes.  Here a file is decoded ONCE by a host thread (Pillow releases the GIL while it
inflates), uploaded as the ``uint8 [H,W,3]`` / integer ``[H,W]`` tensor the augmentation kernel reads (``csrc/augment.hip``) and kept:
the whole decoded Cityscapes training set is 2 975 x (6 MB picture + 4 MB ids as int16) = 30 GB of the 288 GB a MI355X has, so from
the second epoch on a sample costs no host work at all (``dataloader/resident.py`` is the same design with the decode done up front).
A byte budget (``MAS_PICTURE_CACHE_GB``, default 96) bounds the store; beyond it the least recently used entries are dropped.

Formats: pictures -- anything Pillow opens, converted to RGB; label / pseudo-label maps -- 8-bit PNGs or
``.npy`` arrays; superpixel maps -- ``.pkl`` / ``.npy`` pickled dicts with key ``'labels'`` or image files
(``region_cityscapes.py:94-101``).
"""
import collections
import concurrent.futures
import os
import threading

import numpy as np
import torch


def decode_picture(path):
    """``Image.open(path).convert('RGB')`` as a contiguous uint8 [H,W,3] array."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert('RGB'), dtype=np.uint8)             # (a writable, contiguous copy)


def decode_map(path, allow_u8=True):
    """An integer map [H,W] in the narrowest of uint8 / int16 (ids < 32768) / int32 the kernel accepts: superpixel pickles
    (``{'labels': array}``), ``.npy`` arrays, or image files (labelIds / trainIds / pseudo-label PNGs).  ``allow_u8=False`` for
    superpixel ids: the kernel writes uint8 maps back as uint8 (labels), and a crop's pad id ``nseg`` need not fit a byte."""
    ext = path.rsplit('.', 1)[-1].lower()
    if ext in ('pkl', 'npy'):
        data = np.load(path, allow_pickle=True)
        if isinstance(data, np.ndarray) and data.dtype == object:
            data = data.item()
        a = np.asarray(data['labels'] if isinstance(data, dict) else data)
    else:
        from PIL import Image
        with Image.open(path) as im:
            a = np.array(im)
    if a.ndim != 2:
        raise ValueError("%s: expected a 2-D integer map, got shape %s" % (path, a.shape))
    if a.dtype == np.uint8 and allow_u8:
        return np.ascontiguousarray(a)
    lo, hi = int(a.min()), int(a.max())
    if lo < 0:
        raise ValueError("%s: negative ids" % path)
    if hi < 256 and allow_u8:
        return np.ascontiguousarray(a.astype(np.uint8))
    return np.ascontiguousarray(a.astype(np.int16 if hi < 32768 else np.int32))


class PictureStore:
    """path -> device tensor, decoded on first use.  ``prefetch(paths)`` starts the decodes of a batch on the thread pool so that the
    following ``picture`` / ``idmap`` calls find them done (or join them)."""

    def __init__(self, device=None, budget_gb=None, threads=None):
        self.device = device
        self.budget = int(float(budget_gb if budget_gb is not None else os.environ.get("MAS_PICTURE_CACHE_GB", "96")) * (1 << 30))
        self.threads = int(threads if threads is not None else os.environ.get("MAS_DECODE_THREADS", str(min(8, os.cpu_count() or 1))))
        self._pool = None
        self._lock = threading.Lock()
        self._resident = collections.OrderedDict()      # (kind, path) -> device tensor, least recently used first
        self._bytes = 0
        self._pending = {}                              # (kind, path) -> Future of the decoded numpy array
        self.decodes = 0                                # (tests: how many files were decoded)

    def _device(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("a file-backed region dataset produces its samples on the GPU (csrc/augment.hip): no ROCm device is visible")
            self.device = torch.device('cuda', int(os.environ.get("LOCAL_RANK", "0")))
        return self.device

    def _executor(self):
        if self._pool is None:
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, self.threads), thread_name_prefix="mas-decode")
        return self._pool

    @staticmethod
    def _decode(kind, path):
        return decode_picture(path) if kind == 'rgb' else decode_map(path, allow_u8=(kind == 'map'))

    def prefetch(self, items):
        """``items``: iterable of ('rgb' | 'map' | 'ids', path)."""
        with self._lock:
            for key in items:
                if key not in self._resident and key not in self._pending:
                    self._pending[key] = self._executor().submit(self._decode, *key)

    def _get(self, kind, path):
        key = (kind, path)
        with self._lock:
            hit = self._resident.get(key)
            if hit is not None:
                self._resident.move_to_end(key)
                return hit
            fut = self._pending.pop(key, None)
        arr = fut.result() if fut is not None else self._decode(kind, path)
        dev = self._device()
        # the entry outlives the stream it is first used on (a provider prepares batches on a side stream): allocate it on the default
        # stream's pool; the copy from pageable memory has completed when .to() returns
        with torch.cuda.stream(torch.cuda.default_stream(dev)):
            t = torch.from_numpy(arr).to(dev)
        with self._lock:
            self.decodes += 1
            self._resident[key] = t
            self._bytes += t.numel() * t.element_size()
            evict = []
            while self._bytes > self.budget and len(self._resident) > 1:
                _, old = self._resident.popitem(last=False)
                self._bytes -= old.numel() * old.element_size()
                evict.append(old)
        if evict:
            torch.cuda.synchronize(dev)     # (rare) a kernel of any stream may still read an entry that is about to be freed
            del evict
        return t

    def picture(self, path):
        return self._get('rgb', path)

    def labelmap(self, path):
        """A label / pseudo-label map (uint8 when its values fit)."""
        return self._get('map', path)

    def idmap(self, path):
        """A superpixel id map (int16 / int32)."""
        return self._get('ids', path)

    def drop(self, path):
        """Forget a file that was rewritten (a pseudo-label PNG of a new round)."""
        with self._lock:
            for kind in ('rgb', 'map', 'ids'):
                old = self._resident.pop((kind, path), None)
                if old is not None:
                    self._bytes -= old.numel() * old.element_size()

    def resident_bytes(self):
        return self._bytes
