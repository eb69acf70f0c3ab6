"""Multi-scale + flip ensemble of the VOC stage-2 generator -- reference
``trainer/eval_save_cosplbl_prop_includeonehot_voc_ms.py:56-79``: every picture of ``batch['image_list'][0]`` (the
second half of the list is horizontally flipped) goes through ``feat_forward``; features and scores are flipped back,
resized to the original size (bilinear, ``align_corners=False``: torchvision's tensor ``resize`` without antialiasing),
averaged, the features re-normalised over the channels, and handed to the K9 kernels at full resolution.
PNGs go to ``plbl_gen_ms`` (:43)."""
import torch
import torch.nn.functional as F

from . import eval_save_cosplbl_prop_includeonehot_voc


class ActiveTrainer(eval_save_cosplbl_prop_includeonehot_voc.ActiveTrainer):
    def _save_dir(self):
        if self.save_dir is None:
            base = super()._save_dir()
            self.save_dir = base.replace('/plbl_gen', '/plbl_gen_ms', 1)
            import os
            os.makedirs(self.save_dir, exist_ok=True)
        return self.save_dir

    def ensemble(self, image_list, im_size):
        """-> (features [1,Ch,H,W] unit-norm, scores [1,C,H,W]) averaged over the scales / flips."""
        feats = outs = None
        n = len(image_list)
        for idx, img in enumerate(image_list):
            feat, out = self.net.feat_forward(img.to(self.device, dtype=torch.float32)[None])
            if (n - 1) // 2 < idx:                                    # (:64-66) the flipped half
                feat, out = feat.flip(-1), out.flip(-1)
            feat = F.interpolate(feat, size=im_size, mode='bilinear', align_corners=False)
            out = F.interpolate(out, size=im_size, mode='bilinear', align_corners=False)
            feats = feat if feats is None else feats + feat
            outs = out if outs is None else outs + out
        return F.normalize(feats / n, dim=1), outs / n

    def inference_batch(self, batch):
        """One labelled picture: ensemble -> pseudo labels (the reference's loop body, :55-90)."""
        w, h = [int(v) for v in batch['imsizes'][0]]
        feats, outputs = self.ensemble(batch['image_list'][0], (h, w))
        dev = self.device
        return self.pseudo_label_generation(batch['labels'].to(dev), feats, outputs.contiguous(), batch['target'].to(dev),
                                            batch['spmask'].to(dev), batch['spx'].to(dev))

    def pseudo_labels(self, images, labels, targets, spmasks, superpixels):
        raise NotImplementedError("the multi-scale generator consumes batch['image_list']: use inference()")

    def inference(self, loader, prefix=''):
        import numpy as np
        from ..utils.miou import MeanIoU
        meter = MeanIoU(self.num_classes + 1, self.args.ignore_idx)
        meter._before_epoch()
        self.net.eval()
        with torch.no_grad():
            for _ in range(len(loader)):
                batch = next(loader)
                plbl = self.inference_batch(batch)
                meter._after_step({'outputs': plbl, 'targets': batch['labels'].to(self.device, dtype=torch.long)})
                self.after_batch(batch, plbl)
        meter.all_reduce(self.device)
        ious = meter._after_epoch()
        miou = float(np.mean(ious))
        table = ','.join(['%.2f' % miou] + ['%.2f' % v for v in ious])
        print("\n[AL {}-round]: {}\n{}".format(self.selection_iter, prefix, table), flush=True)
        return miou, table
