"""Host-side data contract of the hot path (tensor shapes/dtypes and the active-set bookkeeping).

The reference's ~45 dataset variants, PIL augmentation and on-disk formats are out of scope
(SURVEY.md section 2.1 #6); what the plugins need from the data layer is kept:
``collate_fn`` / ``DataProvider`` (``dataloader/utils.py``) and ``RegionActiveDataset``
(``dataloader/region_active_dataset.py``).
"""
from .region_active_dataset import RegionActiveDataset  # noqa: F401
from .utils import DataProvider, collate_fn  # noqa: F401
