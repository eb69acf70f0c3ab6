"""Stage-2 pseudo labels from a sliding-window ensemble of features and scores (crop 800, stride 2/3) --
reference ``trainer/eval_save_cosplbl_prop_includeonehot_slide.py`` (BASELINE.json config 4: "1024x2048 sliding").

Window features are summed at full resolution on the device (2.1 GB for a Cityscapes image: resident, never copied to
the host), re-normalised over the channels (:72) and handed to the K9 kernels, which accept full-resolution features
(their bilinear interpolation is then the identity)."""
import torch.nn.functional as F

from ..utils.sliding_evaluator_plbl import SlidingEval
from . import eval_save_cosplbl_prop_includeonehot


class ActiveTrainer(eval_save_cosplbl_prop_includeonehot.ActiveTrainer):
    crop_size = 800
    stride_rate = 2 / 3

    def pseudo_labels(self, images, labels, targets, spmasks, superpixels):
        if not hasattr(self, 'evaluator'):
            self.evaluator = SlidingEval(model=self.net, crop_size=self.crop_size, stride_rate=self.stride_rate, device=self.device,
                                         class_number=self.num_classes + 1)
        feats, outputs = self.evaluator(images)
        feats = F.normalize(feats[None], dim=1, p=2)
        return self.pseudo_label_generation(labels, feats, outputs[None].contiguous(), targets, spmasks, superpixels)
