"""VOC form of the production trainer (no "undefined" channel) -- reference
``trainer/active_joint_multi_lossdecomp.py:76-84``: same decomposed objective, targets carry
``num_classes`` columns and all of them are used; selected pixels whose superpixel has no target bit are
skipped (the VOC loss has no assertion, :66-67)."""
from . import active_joint_multi_predignore_lossdecomp as _decomp
from . import active_joint_multi


class ActiveTrainer(active_joint_multi.ActiveTrainer):
    predicts_ignore = False
    get_criterion = _decomp.ActiveTrainer.get_criterion
    losses = _decomp.ActiveTrainer.losses
    train_impl = _decomp.ActiveTrainer.train_impl
