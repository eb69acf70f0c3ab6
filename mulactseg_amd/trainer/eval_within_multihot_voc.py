"""``trainer/eval_within_multihot_voc.py``: the stage-2 evaluation base for PASCAL VOC -- the model has exactly
``num_classes`` (= 21, background included) channels, no extra "undefined" channel (:21)."""
from . import eval_within_multihot


class ActiveTrainer(eval_within_multihot.ActiveTrainer):
    extra_channels = 0
