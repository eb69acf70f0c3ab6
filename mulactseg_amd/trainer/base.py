"""Base trainer -- reference ``trainer/base.py:21-294``.

Differences that are design, not omissions:
* one process per GPU: ``device = cuda:LOCAL_RANK`` (the reference hard-codes ``cuda:0``); under
  ``torch.distributed`` the net is wrapped for gradient all-reduce (RCCL) and evaluation counters are
  all-reduced as integers;
* evaluation uses the fused arg-max + IoU-counter kernel (``utils.miou.LogitsIoU``): one read of the
  logits, no int64 label maps, no per-class host synchronisation;
* ``wandb`` / pandas summary tables are optional (``args.wandb`` is used when present).
"""
import os

import numpy as np
import torch
import torch.optim as optim

from ..dataloader.utils import DataProvider
from ..models import get_model
from ..utils.common import AverageMeter
from ..utils.loss import GroupMultiLabelCE, JointMultiLoss, MultiChoiceCE, MyCrossEntropyLoss
from ..utils.miou import LogitsIoU
from ..utils.scheduler import PolyLR


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


class BaseTrainer(object):
    predicts_ignore = False          # True when the model emits num_classes + 1 channels

    def __init__(self, args, logger):
        self.args = args
        self.logger = logger
        self.model_save_dir = args.model_save_dir
        self.best_iou = 0
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise RuntimeError("the MI355X trainer needs a ROCm device (the reference hard-codes cuda:0, trainer/base.py:27)")
        self.device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(self.device)
        self.am = AverageMeter()

        self.num_classes = args.num_classes
        self.net = self.get_al_model()
        self.net.to(self.device)
        self.ddp = None
        d = _dist()
        if d is not None and d.get_world_size() > 1:
            self.ddp = torch.nn.parallel.DistributedDataParallel(
                self.net, device_ids=[self.local_rank], bucket_cap_mb=getattr(args, 'ddp_bucket_mb', 16),
                gradient_as_bucket_view=True)

        self.get_optim(my_lr=args.train_lr)
        total_itrs = args.finetune_itrs if hasattr(args, "finetune_itrs") else args.total_itrs
        if args.scheduler == 'poly':
            self.scheduler = PolyLR(self.optimizer, total_itrs, power=args.power, min_lr=args.min_lr)
        elif args.scheduler != 'none':
            raise NotImplementedError
        self.get_criterion()

    # -- construction hooks (overridden by the plugins) -------------------------------------------
    def get_al_model(self):
        a = self.args
        return get_model(model=a.model, num_classes=self.num_classes, output_stride=a.output_stride,
                         separable_conv=a.separable_conv,
                         pretrained_backbone=getattr(a, 'pretrained_backbone', True))

    def get_optim(self, my_lr):
        groups = [{'params': list(self.net.backbone.parameters()), 'lr': my_lr},
                  {'params': list(self.net.classifier.parameters()), 'lr': self.args.cls_lr_scale * my_lr}]
        if self.args.optimizer == 'adamw':
            # same update rule as the reference's optim.AdamW (trainer/base.py:64-66); on the GPU the single-kernel form
            on_gpu = any(p.is_cuda for g in groups for p in g['params'])
            if on_gpu and os.environ.get("MAS_ADAMW", "own") == "own":
                from ..utils.optim import FusedAdamW        # one launch over all parameters (csrc/optim.hip)
                self.optimizer = FusedAdamW(params=groups, lr=my_lr, weight_decay=self.args.weight_decay)
            else:
                self.optimizer = optim.AdamW(params=groups, lr=my_lr, weight_decay=self.args.weight_decay, fused=on_gpu)
        elif self.args.optimizer == 'sgd':
            self.optimizer = optim.SGD(params=groups, lr=my_lr, momentum=0.9, weight_decay=self.args.weight_decay)
        else:
            raise NotImplementedError

    def get_criterion(self):
        """The in-scope ``--loss_type`` values of ``trainer/base.py:73-112`` (the production trainers
        override this; the reference's own 'joint_multi_loss' branch cannot be constructed, Appendix D)."""
        a = self.args
        if a.loss_type == 'cross_entropy':
            self.loss_fun = MyCrossEntropyLoss(ignore_index=a.ignore_idx, reduction='mean', temperature=a.ce_temp)
        elif a.loss_type == 'multi_choice_ce':
            self.loss_fun = MultiChoiceCE(num_class=self.num_classes, temperature=a.multi_ce_temp)
        elif a.loss_type == 'group_multi_label_ce':
            self.loss_fun = GroupMultiLabelCE(args=a, num_class=self.num_classes, num_superpixel=a.nseg, temperature=a.group_ce_temp)
        elif a.loss_type == 'joint_multi_loss':
            self.loss_fun = JointMultiLoss(
                GroupMultiLabelCE(args=a, num_class=self.num_classes, num_superpixel=a.nseg, temperature=a.group_ce_temp),
                MultiChoiceCE(num_class=self.num_classes, temperature=a.multi_ce_temp))
        else:
            raise NotImplementedError("loss_type %r is outside the hot path" % a.loss_type)

    # -- loaders ------------------------------------------------------------------------------------
    def loader_seed(self):
        """Seed of the training loader's PRIVATE generators: a function of (--seed, rank, AL round), so that data-parallel
        ranks draw different batches and crops (N GPUs x batch 4 = one batch of 4N distinct samples) while the process-global
        ``random`` / numpy / torch generators -- which ``my_random`` and the selectors read -- stay identical on all ranks."""
        d = _dist()
        rank = d.get_rank() if d is not None else 0
        return (int(getattr(self.args, 'seed', 0)) * 1000003 + rank) * 1009 + int(getattr(self, 'selection_iter', 0))

    def get_trainloader(self, dataset):
        import random
        seed = self.loader_seed()
        if getattr(dataset, 'device_resident', False):      # samples are produced on the device (dataloader/resident.py)
            from ..dataloader.utils import ResidentProvider
            if hasattr(getattr(dataset, 'transform', None), 'rng'):
                dataset.transform.rng = random.Random(seed ^ 0x5bd1e995)           # crops / scales / flips
            return ResidentProvider(dataset, batch_size=self.args.train_batch_size, drop_last=True, shuffle=True,
                                    rng=random.Random(seed))
        gen = torch.Generator()
        gen.manual_seed(seed)               # epoch permutations and the workers' base seed (hence their `random` streams)
        return DataProvider(dataset=dataset, batch_size=self.args.train_batch_size, shuffle=True,
                            num_workers=self.args.num_workers, pin_memory=True, drop_last=True, generator=gen)

    def get_valloader(self, dataset):
        # one process per GPU: every rank evaluates its round-robin share; inference() sums the IoU counters over ranks
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dataset = torch.utils.data.Subset(dataset, range(dist.get_rank(), len(dataset), dist.get_world_size()))
        if getattr(getattr(dataset, 'dataset', dataset), 'device_resident', False):
            from ..dataloader.utils import ResidentProvider
            return ResidentProvider(dataset, batch_size=self.args.val_batch_size, drop_last=False, shuffle=False)
        return DataProvider(dataset=dataset, batch_size=self.args.val_batch_size, shuffle=False,
                            num_workers=self.args.val_num_workers, pin_memory=True, drop_last=False)

    def train(self):
        raise NotImplementedError

    def train_impl(self, total_itrs, val_period):
        raise NotImplementedError

    def forward_train(self, images, **kw):
        return (self.ddp or self.net)(images, **kw)

    # -- the stream-K convolutions' error word (csrc/conv_sk.hip: a finisher that gave up waiting poisons its tile and says so) --
    def stream_k_flag(self):
        """[] or [0-dim device tensor != 0 when a stream-K launch of ANY rank's device has given up since the last clear]: no
        host synchronisation -- the trainers read it together with the loss.  Under DDP the word is MAX-all-reduced, so every rank
        sees the same flag at the same step and they raise together (one rank leaving while the others enter the gradient
        all-reduce would hang them until the RCCL timeout)."""
        from .. import ops
        if not ops.sk_split_default():              # whole-tile plan (the default under DDP): no hand-off, nothing can give up
            return []
        words = ops.conv_sk_error_words(self.device)
        if words is None:
            return []
        flag = words.max().to(torch.int32)
        if getattr(self, 'ddp', None) is not None:
            import torch.distributed as dist
            flag = flag.reshape(1)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            flag = flag.reshape(())
        return [flag]

    def guard_optimizer_step(self):
        """Between backward() and optimizer.step(): a give-up inside THIS step's forward or backward pass must not reach the
        parameters.  The host does not wait for the backward pass (the flag travels to it with the NEXT step's loss), so the guard
        is on the device: the fused AdamW kernel takes a `found_inf` scalar (the GradScaler protocol, torch/optim/adam.py) and leaves
        parameters, moments and step counts untouched when it is non-zero.  No-op for optimizers without the fused kernel."""
        opt = self.optimizer
        fused = any(g.get('fused') for g in opt.param_groups)
        flag = self.stream_k_flag() if fused else []
        if flag:
            opt.found_inf = (flag[0] != 0).to(torch.float32)
        elif hasattr(opt, 'found_inf'):
            del opt.found_inf

    def raise_stream_k(self):
        from .. import ops
        ops.conv_sk_clear_error(self.device)
        raise ops.StreamKGaveUp(
            "a stream-K convolution gave up waiting for a workgroup of its own launch (mas_conv_sk error word): the GPU is shared with "
            "something that held its CUs for seconds; the activations / gradients of that step are poisoned (NaN).  The optimizer "
            "update of that step was skipped on the device (guard_optimizer_step), BatchNorm running statistics of the layers behind "
            "the poisoned tile are not: reload the last checkpoint, or run with MAS_SK_SPLIT=off (whole-tile plan, no hand-off)")

    def check_stream_k(self):
        """Synchronous form (end of a training round, tests)."""
        flag = self.stream_k_flag()
        if flag and int(flag[0].item()) != 0:
            self.raise_stream_k()

    # -- evaluation ---------------------------------------------------------------------------------
    def inference(self, loader, prefix=''):
        """mIoU over a loader -- ``trainer/base.py:139-175`` / ``active_joint_multi_predignore.py:175-215``."""
        meter = LogitsIoU(self.num_classes, self.args.ignore_idx)
        meter._before_epoch()
        self.net.eval()
        with torch.no_grad():
            for _ in range(len(loader)):
                batch = next(loader)
                images = batch['images'].to(self.device, dtype=torch.float32)
                labels = batch['labels'].to(self.device, dtype=torch.long)
                meter.step(self.net(images).detach(), labels)
        meter.all_reduce(self.device)
        ious = meter.ious()
        miou = np.mean(ious)
        cells = ['%.2f' % miou] + ['%.2f' % v for v in ious]
        if self.predicts_ignore:
            cells.append('%.2f' % meter.ignore_iou())
        table = ','.join(cells)
        print("\n[AL {}-round]: {}\n{}".format(getattr(self, 'selection_iter', 0), prefix, table), flush=True)
        return miou, table

    def _wandb_log(self, payload, step):
        log = getattr(getattr(self.args, 'wandb', None), 'log', None)
        if log is not None:
            log(payload, step=step)

    def validate(self, trainiter=None, prefix=''):
        miou, table = self.inference(loader=self.val_dataset_loader, prefix='validation')
        self.logger.info('[Validation Result]')
        self.logger.info('%s' % table)
        if self.best_iou < miou:
            self.best_iou = miou
            self.save_checkpoint()
        self.logger.info('Current val miou is %.3f %%, while the best val miou is %.3f %%' % (miou, self.best_iou))
        step = trainiter + int(self.args.finetune_itrs) * (self.selection_iter - 1)
        self._wandb_log({'{}val-miou'.format(prefix): miou, '{}val-best-miou'.format(prefix): self.best_iou,
                         '{}selection_iter'.format(prefix): self.selection_iter}, step + 1)
        return table

    def eval(self, selection_iter):
        miou, table = self.inference(loader=self.eval_dataset_loader, prefix='evaluation')
        self.logger.info('[Evaluation Result]')
        self.logger.info('%s' % table)
        self.logger.info('Current eval miou is %.3f %%' % miou)
        self._wandb_log({'eval-miou': miou, 'selection_iter': selection_iter}, int(self.args.finetune_itrs) * selection_iter)
        tab = getattr(self.args, 'wandb_iou_table', None)
        if tab is not None:
            tab.loc[0, 'round_v_miou'] = "{}{:.2f},".format(tab.loc[0]['round_v_miou'], miou)
            tab.loc[0, "round-{}".format(selection_iter)] = table
        return table

    # -- checkpoints (``trainer/base.py:281-294``) -----------------------------------------------------
    def save_checkpoint(self):
        d = _dist()
        if d is None or d.get_rank() == 0:
            torch.save({'model_state_dict': self.net.state_dict(), 'opt_state_dict': self.optimizer.state_dict()},
                       self.checkpoint_file)

    def load_checkpoint(self, fname, load_optimizer=False):
        checkpoint = torch.load(fname, map_location=self.device)
        self.net.load_state_dict(checkpoint['model_state_dict'])
        if load_optimizer is True:
            self.optimizer.load_state_dict(checkpoint['opt_state_dict'])
