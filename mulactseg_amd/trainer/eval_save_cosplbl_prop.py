"""Stage-2 pseudo-label generation: cosine prototypes + one-ring propagation over the multi-hot superpixels, saved
as uint8 PNGs -- reference ``trainer/eval_save_cosplbl_prop.py:20-314``.

The reference materialises ``feat_forward``'s 256-channel full-resolution features (2.1 GB per Cityscapes image) and
walks the selected superpixels in Python with a CPU dilation each; here the quarter-resolution features go straight
into the K9 kernels (``ops.stage2_pseudo_labels``, ``csrc/stage2.hip``)."""
import os
import queue
import threading

import numpy as np
import torch

from .. import ops
from ..utils.miou import MeanIoU
from . import eval_within_multihot


class ActiveTrainer(eval_within_multihot.ActiveTrainer):
    include_onehot = False

    def __init__(self, args, logger, selection_iter):
        super().__init__(args, logger, selection_iter)
        assert args.val_batch_size == 1
        self.save_dir = None

    def _save_dir(self):
        """``<dir of init_checkpoint>/plbl_gen[_<plbl_type>]/round_RR`` (:33-39)."""
        if self.save_dir is None:
            ckpt = self.args.init_checkpoint
            rnd = ckpt.split('/')[-1][-6:-4]
            base = '/'.join(ckpt.split('/')[:-1])
            ptype = getattr(self.args, 'plbl_type', None)
            self.save_dir = '{}/plbl_gen{}/round_{}'.format(base, '' if ptype is None else '_' + ptype, rnd)
            os.makedirs(self.save_dir, exist_ok=True)
        return self.save_dir

    def pseudo_labels(self, images, labels, targets, spmasks, superpixels):
        feats, outputs = self.net.feat_forward_lowres(images)
        return self.pseudo_label_generation(labels, feats, outputs, targets, spmasks, superpixels)

    def pseudo_label_generation(self, labels, feats, inputs, targets, spmasks, superpixels):
        """Same signature as the reference (:121); ``feats`` may be the quarter-resolution map."""
        return ops.stage2_pseudo_labels(feats.contiguous(), inputs.contiguous(), targets.contiguous(), spmasks.contiguous(),
                                        superpixels.contiguous(), include_onehot=self.include_onehot)

    def inference(self, loader, prefix=''):
        """The loop of ``eval_within_multihot.inference`` (:55-71) with the pictures dealt to ``MAS_STAGE2_WORKERS`` (default 4) threads,
        each on its own HIP stream: a picture's generation has host waits in it (the prototype list is a ``nonzero``, the label map
        goes to the host, the PNG is encoded there), so one picture at a time leaves the GPU idle for 60 % of the loop -- 33 ms per
        1024 x 2048 picture for 13 ms of kernels; 10.9 ms with four threads (``tools/stage2_loop_probe.py``).  The loader is read by the calling thread, in order; a PNG is a per-picture file and
        the IoU counters are integer sums, so neither depends on which thread took which picture.  The first picture runs on the
        caller's stream (everything the model derives lazily from its weights is built there).  Subclasses that replace
        ``pseudo_labels`` (the sliding-window form keeps state between calls) run the one-thread loop."""
        workers = int(os.environ.get("MAS_STAGE2_WORKERS", "4"))
        dev = torch.device(self.device)
        if workers <= 1 or dev.type != 'cuda' or type(self).pseudo_labels is not ActiveTrainer.pseudo_labels or len(loader) < 2:
            return super().inference(loader, prefix)
        meter = MeanIoU(self.num_classes + 1, self.args.ignore_idx)
        meter._before_epoch()
        self.net.eval()
        self._save_dir()
        jobs, errors = queue.Queue(maxsize=2 * workers), []

        def one(batch):
            images, labels, superpixels, spmasks, targets = self._batch(batch)
            plbl = self.pseudo_labels(images, labels, targets, spmasks, superpixels)
            meter._after_step({'outputs': plbl, 'targets': labels})
            self.after_batch(batch, plbl)               # (ends with the label map on the host: this stream has drained)

        def work(stream):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(stream), torch.no_grad():
                while True:
                    item = jobs.get()
                    if item is None:
                        return
                    if errors:
                        continue
                    batch, ready = item
                    try:
                        stream.wait_event(ready)
                        one(batch)
                        stream.synchronize()
                    except BaseException as e:          # noqa: BLE001 (re-raised by the calling thread)
                        errors.append(e)
        threads = []
        with torch.no_grad():
            for k in range(len(loader)):
                batch = next(loader)
                if k == 0:
                    one(batch)
                    main = torch.cuda.current_stream(dev)
                    for _ in range(workers):
                        st = torch.cuda.Stream(device=dev)
                        st.wait_stream(main)
                        threads.append(threading.Thread(target=work, args=(st,), daemon=True))
                        threads[-1].start()
                    continue
                if errors:
                    break
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(dev))    # the batch was made (or copied) by work queued so far on this stream
                jobs.put((batch, ready))
        for _ in threads:
            jobs.put(None)
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        meter.all_reduce(self.device)
        ious = meter._after_epoch()
        miou = np.mean(ious)
        table = ','.join(['%.2f' % miou] + ['%.2f' % v for v in ious])
        print("\n[AL {}-round]: {}\n{}".format(self.selection_iter, prefix, table), flush=True)
        return miou, table

    def after_batch(self, batch, plbl):
        from PIL import Image
        fname = batch['fnames'][0][1]
        lbl_id = fname.split('/')[-1].split('.')[0]
        Image.fromarray(plbl[0].cpu().numpy().astype('uint8')).save("{}/{}.png".format(self._save_dir(), lbl_id))
