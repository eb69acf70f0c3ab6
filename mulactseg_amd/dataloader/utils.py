"""``collate_fn`` and ``DataProvider`` with the behaviour of the reference's ``dataloader/utils.py:10-62``
(the reference's own ``DataProvider`` calls ``iterator.next()``, which no longer exists in torch 2)."""
import numpy as np
import torch
from torch.utils.data import DataLoader

_STACKED = ('images', 'image_weak', 'spx', 'spx_weak', 'spmask', 'spmask_weak', 'labels', 'spx_small',
            'spx_small_weak', 'target', 'nseg_list')
_LISTED = ('image_list', 'fnames', 'imsizes')


def collate_fn(inputs):
    """Stack tensor-like entries, keep file names / sizes as lists -- ``dataloader/utils.py:10-25``."""
    batch = {}
    for key in inputs[0].keys():
        vals = [item[key] for item in inputs]
        if key in _STACKED:
            if isinstance(vals[0], np.ndarray):
                vals = [torch.from_numpy(v) for v in vals]
            batch[key] = torch.stack(vals)
        elif key in _LISTED or 'mseg_' in key:
            batch[key] = vals
        else:
            raise NotImplementedError("collate_fn: unknown batch key %r" % key)
    return batch


class DataProvider:
    """Endless iterator over a DataLoader (epoch counter, restart on exhaustion) --
    ``dataloader/utils.py:28-62``."""

    def __init__(self, dataset, batch_size, num_workers, drop_last, shuffle, pin_memory, generator=None):
        self.dataset = dataset
        self.iteration = 0
        self.epoch = 0
        self.batch_size, self.num_workers = batch_size, num_workers
        self.drop_last, self.shuffle, self.pin_memory = drop_last, shuffle, pin_memory
        self.dataloader = DataLoader(dataset, batch_size=batch_size, collate_fn=collate_fn, shuffle=shuffle,
                                     num_workers=num_workers, drop_last=drop_last, pin_memory=pin_memory, generator=generator)
        self.dataiter = iter(self.dataloader)

    def __len__(self):
        return len(self.dataloader)

    def __next__(self):
        try:
            batch = next(self.dataiter)
        except StopIteration:
            self.epoch += 1
            self.dataiter = iter(self.dataloader)
            batch = next(self.dataiter)
        self.iteration += 1
        return batch


class ResidentProvider:
    """``DataProvider`` for datasets whose samples are produced ON the device (``dataloader/resident.py``): no worker
    processes, no pinned staging -- batches are stacked device tensors.  Same endless-iterator surface
    (``len``, ``next``, ``epoch``, ``iteration``); shuffling draws a fresh permutation per epoch from ``rng``
    (a ``random.Random`` or the ``random`` module), as ``DataLoader(shuffle=True)`` does per epoch.

    ``prefetch``: the NEXT batch is prepared on a side HIP stream while the caller trains on the current one (the
    augmentation kernels are ~0.4 ms per sample; the order of the random draws -- and therefore every batch -- is the
    same as without prefetching)."""

    def __init__(self, dataset, batch_size, drop_last=True, shuffle=True, rng=None, prefetch=True):
        import random as _random
        self.dataset, self.batch_size, self.drop_last, self.shuffle = dataset, int(batch_size), drop_last, shuffle
        # a private generator by default: the batch prepared ahead (and possibly never consumed) must not advance the
        # process-global ``random`` that ``my_random`` and expand_training_set's callers read
        self.rng = rng if rng is not None else _random.Random(_random.getrandbits(64) if shuffle else 0)
        self.iteration = 0
        self.epoch = 0
        self._order, self._pos = [], 0
        self._epochs_drawn = 0
        self._new_epoch()
        self._side = torch.cuda.Stream() if (prefetch and torch.cuda.is_available()) else None
        self._pending = None

    def _new_epoch(self):
        self._order = list(range(len(self.dataset)))
        if self.shuffle:
            self.rng.shuffle(self._order)
        self._pos = 0

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _draw(self):
        """Indices of the next batch (and whether it opens a new epoch)."""
        new_epoch = False
        if self._pos + (self.batch_size if self.drop_last else 1) > len(self._order):
            self._new_epoch()
            new_epoch = True
        idx = self._order[self._pos:self._pos + self.batch_size]
        self._pos += len(idx)
        return idx, new_epoch

    def _prepare(self):
        idx, new_epoch = self._draw()
        # a file-backed dataset (picture_store.PictureStore) starts the decodes of the whole batch on its host threads first
        base, sub = self.dataset, None
        if isinstance(base, torch.utils.data.Subset):
            base, sub = base.dataset, base.indices
        if hasattr(base, 'prefetch'):
            base.prefetch([i if sub is None else sub[i] for i in idx])
        if self._side is None:
            return collate_fn([self.dataset[i] for i in idx]), None, new_epoch
        self._side.wait_stream(torch.cuda.current_stream())       # resident tensors written on the main stream are visible
        with torch.cuda.stream(self._side):
            batch = collate_fn([self.dataset[i] for i in idx])
            ev = torch.cuda.Event()
            ev.record(self._side)
        return batch, ev, new_epoch

    def __next__(self):
        if len(self) == 0:
            raise StopIteration("dataset smaller than one batch")
        if self._pending is None:
            self._pending = self._prepare()
        batch, ev, new_epoch = self._pending
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for v in batch.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        self.epoch += int(new_epoch)
        self.iteration += 1
        self._pending = self._prepare() if self._side is not None else None     # overlaps with the caller's training step
        return batch
