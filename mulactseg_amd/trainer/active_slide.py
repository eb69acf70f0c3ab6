"""``trainer/active_slide.py:9-55``: the plain trainer evaluated on the full-resolution "slide" dataset
(``dataloader.get_slide_dataset``: Cityscapes resized to 1024x2048, no crop) with whole-image forwards."""
from ..dataloader import get_slide_dataset
from . import active


class ActiveTrainer(active.ActiveTrainer):
    def __init__(self, args, logger, selection_iter):
        super().__init__(args, logger, selection_iter)
        eval_dataset = get_slide_dataset(name=self.args.val_dataset, data_root=self.args.val_data_dir,
                                         datalist=self.args.val_datalist, imageset='eval')
        self.eval_dataset_loader = self.get_valloader(eval_dataset)
