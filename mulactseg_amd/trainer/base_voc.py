"""``trainer/base_voc.py``: the reference's copy of ``trainer/base.py`` that imports ``AverageMeter`` from
``utils/common_voc`` (the only difference, base_voc.py:19); the meter is the same class here."""
from .base import BaseTrainer  # noqa: F401
