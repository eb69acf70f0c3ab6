"""Joint multi-label trainer that additionally predicts the "undefined" class (Cityscapes) --
reference ``trainer/active_joint_multi_predignore.py:130-215``: the model emits ``num_classes + 1``
channels, the losses use every target column, the ImageNet checkpoint loads without its classifier."""
import torch

from ..models import get_model
from ..utils.loss import GroupMultiLabelCE_, MultiChoiceCE_
from . import active_joint_multi


class ActiveTrainer(active_joint_multi.ActiveTrainer):
    predicts_ignore = True

    def get_criterion(self):
        a = self.args
        self.group_multi_loss = GroupMultiLabelCE_(args=a, num_class=self.num_classes, num_superpixel=a.nseg, temperature=a.group_ce_temp)
        self.multi_pos_loss = MultiChoiceCE_(num_class=self.num_classes, temperature=a.multi_ce_temp)

    def get_al_model(self):
        a = self.args
        return get_model(model=a.model, num_classes=self.num_classes + 1, output_stride=a.output_stride,
                         separable_conv=a.separable_conv, pretrained_backbone=getattr(a, 'pretrained_backbone', True))

    def load_checkpoint(self, fname, load_optimizer=False):
        checkpoint = torch.load(fname, map_location=self.device)
        if 'imagenet_pretrained' in fname:       # class count changed: drop the classifier (:156-171)
            for key in ('classifier.final.weight', 'classifier.final.bias', 'classifier.proxy'):
                checkpoint['model_state_dict'].pop(key, None)
        self.net.load_state_dict(checkpoint['model_state_dict'], strict=False)
        if load_optimizer is True:
            self.optimizer.load_state_dict(checkpoint['opt_state_dict'])
