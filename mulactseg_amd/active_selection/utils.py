"""Pool loader for the acquisition passes -- reference ``active_selection/utils.py:47-57``
(plain sequential DataLoader, ``shuffle=False``).  ``indices`` restricts it to this rank's shard."""
import torch

from ..dataloader.utils import collate_fn


class _DecodeAhead(torch.utils.data.Dataset):
    """A file-backed pool (``dataloader/picture_store.py``) read front to back: asking for sample k starts the host decodes of the
    next ``window`` samples on the store's threads, so the first pass over a pool is bound by ``threads`` decoders, not by one."""

    def __init__(self, pool_set, order, window):
        self.pool_set, self.order, self.window, self.started = pool_set, list(order), int(window), 0

    def __len__(self):
        return len(self.order)

    def __getitem__(self, k):
        hi = min(len(self.order), k + 1 + self.window)
        if hi > self.started:
            self.pool_set.prefetch(self.order[max(self.started, k):hi])
            self.started = hi
        return self.pool_set[self.order[k]]


def get_al_loader(trainer, pool_set, batch_size, num_workers, indices=None):
    dataset = pool_set if indices is None else torch.utils.data.Subset(pool_set, list(indices))
    if hasattr(pool_set, 'prefetch'):
        dataset = _DecodeAhead(pool_set, range(len(pool_set)) if indices is None else indices, 2 * max(int(batch_size), 8))
    resident = getattr(pool_set, 'device_resident', False)      # samples are device tensors already: no workers, nothing to pin
    loader = torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False,
                                         num_workers=0 if resident else num_workers, collate_fn=collate_fn,
                                         pin_memory=torch.cuda.is_available() and not resident, sampler=None)
    return loader, 0
