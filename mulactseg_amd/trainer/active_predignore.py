"""Stage-2 trainer: plain temperature CE on pseudo labels with the "undefined" channel --
reference ``trainer/active_predignore.py:12-94``."""
import torch

from . import active, active_joint_multi_predignore


class ActiveTrainer(active.ActiveTrainer):
    predicts_ignore = True
    get_al_model = active_joint_multi_predignore.ActiveTrainer.get_al_model
    load_checkpoint = active_joint_multi_predignore.ActiveTrainer.load_checkpoint

    def __init__(self, args, logger, selection_iter):
        super().__init__(args, logger, selection_iter)
        self.target_dtype = torch.long
