"""Pool loader for the acquisition passes -- reference ``active_selection/utils.py:47-57``
(plain sequential DataLoader, ``shuffle=False``).  ``indices`` restricts it to this rank's shard."""
import torch

from ..dataloader.utils import collate_fn


def get_al_loader(trainer, pool_set, batch_size, num_workers, indices=None):
    dataset = pool_set if indices is None else torch.utils.data.Subset(pool_set, list(indices))
    resident = getattr(pool_set, 'device_resident', False)      # samples are device tensors already: no workers, nothing to pin
    loader = torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False,
                                         num_workers=0 if resident else num_workers, collate_fn=collate_fn,
                                         pin_memory=torch.cuda.is_available() and not resident, sampler=None)
    return loader, 0
