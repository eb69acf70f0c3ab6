"""Active-learning trainer -- reference ``trainer/active.py:10-104``."""
import os

import torch

from ..dataloader import get_dataset
from .base import BaseTrainer


class ActiveTrainer(BaseTrainer):
    def __init__(self, args, logger, selection_iter):
        self.selection_iter = selection_iter
        super().__init__(args, logger)
        self.target_dtype = torch.uint8 if getattr(args, 'or_labeling', False) else torch.long
        if 'oracle' in getattr(args, 'loader', ''):
            self.target_dtype = torch.long
        val_dataset = get_dataset(args, name=args.val_dataset, data_root=args.val_data_dir,
                                  datalist=args.val_datalist, imageset='val')
        eval_dataset = get_dataset(args, name=args.val_dataset, data_root=args.val_data_dir,
                                   datalist=args.val_datalist, imageset='eval')
        self.val_dataset_loader = self.get_valloader(val_dataset)
        self.eval_dataset_loader = self.get_valloader(eval_dataset)

    def get_optim(self, my_lr):
        if getattr(self.args, 'adaptive_train_lr', False):
            my_lr = self.args.train_lr * self.selection_iter
        super().get_optim(my_lr=my_lr)

    def train(self, active_set, fname=None):
        train_dataset = active_set.get_trainset()
        if fname is None:
            self.checkpoint_file = os.path.join(self.model_save_dir, 'checkpoint%02d.tar' % active_set.selection_iter)
        else:
            self.checkpoint_file = fname
        self.train_dataset_loader = self.get_trainloader(train_dataset)
        self.train_impl(int(self.args.finetune_itrs), int(self.args.val_period))

    def log_validation(self, iteration, val_period):
        if iteration % val_period == (val_period - 1) and iteration > self.args.val_start:
            self.logger.info('**** EVAL ITERATION %06d ****' % iteration)
            self.validate(trainiter=iteration)
            self.net.train()

    def log_training(self, iteration, pbar, total_itrs):
        if iteration % self.args.log_period == (self.args.log_period - 1):
            step = iteration + total_itrs * (self.selection_iter - 1)
            payload = {'learning-rate cls': self.optimizer.param_groups[-1]['lr']}
            payload.update({k: self.am.pop(k) for k in list(self.am.get_whole_data())})
            self._wandb_log(payload, step)

    def train_impl(self, total_itrs, val_period):
        """Plain CE training step (``trainer/active.py:73-104``)."""
        self.net.train()
        for iteration in range(total_itrs):
            batch = next(self.train_dataset_loader)
            images = batch['images'].to(self.device, dtype=torch.float32)
            labels = batch['labels'].to(self.device, dtype=self.target_dtype)
            self.optimizer.zero_grad()
            net = self.ddp or self.net
            if (hasattr(self.loss_fun, 'forward_lowres') and getattr(self.net, 'lowres_logits', False) and images.is_cuda
                    and getattr(self.args, 'lowres_loss', True) and labels.dim() == 3):
                # quarter-resolution logits out of the model, the final x4 bilinear upsampling inside the loss scans (a-11)
                loss = self.loss_fun.forward_lowres(net(images, lowres=True), images.shape[-2:], labels)
            else:
                loss = self.loss_fun(self.forward_train(images), labels)
            bad = torch.isnan(loss.detach()).to(torch.int32)
            if self.ddp is not None:            # the skip must be taken by every rank or by none (gradient all-reduce)
                import torch.distributed as dist
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            # (the stream-K error word rides on the same host read: this forward pass and the previous step's backward pass)
            host = torch.stack([bad.reshape(())] + [f.to(torch.int32).reshape(()) for f in self.stream_k_flag()]).tolist()
            if len(host) > 1 and host[1] != 0:
                self.raise_stream_k()
            ok = not host[0]
            if ok:
                loss.backward()
                self.guard_optimizer_step()         # (device-side: a give-up in this step's passes skips the parameter update)
                self.optimizer.step()
            if self.args.scheduler == 'poly':
                self.scheduler.step()
            if ok:
                self.am.add({'train-loss': loss.detach().cpu().item()})
            self.log_training(iteration, None, total_itrs)
            self.log_validation(iteration, val_period)
        self.check_stream_k()
