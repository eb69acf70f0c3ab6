"""Joint multi-label trainer -- reference ``trainer/active_joint_multi.py:8-76``:
``loss = coeff * merged_positive_CE + group_loss``; the step is skipped when the loss is exactly 0 and
a NaN raises (``check_loss_sanity`` :23-29)."""
import torch

from ..utils.loss import GroupMultiLabelCE, MultiChoiceCE
from . import active


class _HostProbe:
    """Device scalars to the host without draining the training stream: the values are packed on the current stream, an
    event marks that point, and a side stream copies them into pinned memory; ``read`` waits for THAT copy only -- not for
    whatever was queued on the training stream afterwards (the backward pass)."""

    def __init__(self, device, slots=4, width=8):
        self.stream = torch.cuda.Stream(device)
        self.bufs = [torch.empty(width, dtype=torch.float32).pin_memory() for _ in range(slots)]
        self.k = 0

    def submit(self, values):
        packed = torch.stack([v.detach().float().reshape(()) for v in values])
        cur = torch.cuda.current_stream(packed.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        buf = self.bufs[self.k % len(self.bufs)]
        self.k += 1
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            buf[:packed.numel()].copy_(packed, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        packed.record_stream(self.stream)
        return done, buf, packed.numel()

    @staticmethod
    def read(handle):
        done, buf, n = handle
        done.synchronize()
        return buf[:n].tolist()


class ActiveTrainer(active.ActiveTrainer):
    def get_criterion(self):
        a = self.args
        self.group_multi_loss = GroupMultiLabelCE(args=a, num_class=self.num_classes, num_superpixel=a.nseg, temperature=a.group_ce_temp)
        self.multi_pos_loss = MultiChoiceCE(num_class=self.num_classes, temperature=a.multi_ce_temp)

    def zero_if_nan(self, loss):
        return 0 if torch.isnan(loss) else loss

    def check_loss_sanity(self, loss):
        """Reference semantics (``active_joint_multi.py:31-37``): a zero loss (no selected pixel in the batch) skips
        the step, NaN raises.  One device->host read instead of the reference's two."""
        v = float(loss.detach())
        if v == 0:
            return False
        if v != v:
            raise ValueError("NaN loss")
        return True

    def _probe(self):
        p = getattr(self, '_host_probe', None)
        if p is None:
            p = self._host_probe = _HostProbe(self.device)
        return p

    def update(self, loss):
        """Reference semantics (``active_joint_multi.py:31-42``): a zero loss skips the optimizer step, NaN raises.  On the GPU
        the backward pass is LAUNCHED before the host looks at the loss value (read through a side stream, so the host waits
        for the forward pass only): the device never idles behind a host round trip between forward and backward.  A zero
        loss then has an all-zero gradient, and skipping ``optimizer.step()`` leaves parameters and optimizer state exactly
        as skipping the backward pass does."""
        # DDP averages gradients over ranks; every partial-label loss module already returns the GLOBAL-batch objective
        # (normalisers all-reduced, identical value on every rank -- so the skip is taken by all ranks or by none), hence
        # scale by the world size to get its exact gradient.
        scale = 1
        if self.ddp is not None and getattr(self, 'loss_is_global', True):
            # (trainers whose criterion does NOT all-reduce its normalisers set loss_is_global = False: their loss is the local
            # objective and DistributedDataParallel's gradient average is already what the reference's DataParallel computes)
            import torch.distributed as dist
            scale = dist.get_world_size()
        if loss.is_cuda:
            # the error words of the stream-K convolutions travel with the loss: this forward pass and the PREVIOUS step's backward
            # pass are covered here, the last backward pass of a round by check_stream_k() at the end of train_impl
            handle = self._probe().submit([loss] + self.stream_k_flag())
            (loss * scale if scale != 1 else loss).backward()
            self.guard_optimizer_step()             # (device-side: a give-up in this step's passes skips the parameter update)
            host = _HostProbe.read(handle)
            v = host[0]
            if len(host) > 1 and host[1] != 0:
                self.raise_stream_k()
            if v != v:
                raise ValueError("NaN loss")
            if v == 0:
                self.optimizer.zero_grad()
            else:
                self.optimizer.step()
        elif self.check_loss_sanity(loss):
            (loss * scale if scale != 1 else loss).backward()
            self.optimizer.step()
        if self.args.scheduler == 'poly':
            self.scheduler.step()

    def update_average_meter(self, values):
        """All meters from ONE device->host transfer (the reference syncs twice per value), and on the GPU a deferred one: the
        values of step t are read when step t + 1 reports (or before anything pops the meters), so reporting never drains the
        stream."""
        self.flush_meters()
        keys = list(values)
        if all(values[k].is_cuda for k in keys):
            self._pending_meters = (keys, self._probe().submit([values[k] for k in keys]))
            return
        self._add_meters(keys, torch.stack([values[k].detach().float().reshape(()) for k in keys]).cpu().tolist())

    def _add_meters(self, keys, host):
        for key, v in zip(keys, host):
            if v != v:
                raise ValueError("NaN loss")
            if v != 0:
                self.am.add({key: v})

    def flush_meters(self):
        pending = getattr(self, '_pending_meters', None)
        if pending is not None:
            self._pending_meters = None
            self._add_meters(pending[0], _HostProbe.read(pending[1]))

    def log_training(self, iteration, pbar, total_itrs):
        if iteration % self.args.log_period == (self.args.log_period - 1):
            self.flush_meters()                 # the meters about to be popped include this step
        super().log_training(iteration, pbar, total_itrs)

    def _batch(self):
        batch = next(self.train_dataset_loader)
        images = batch['images'].to(self.device, dtype=torch.float32)
        labels = batch['labels'].to(self.device, dtype=self.target_dtype)
        return images, labels, batch['spx'].to(self.device), batch['spmask'].to(self.device)

    def train_impl(self, total_itrs, val_period):
        self.net.train()
        for iteration in range(total_itrs):
            images, labels, superpixels, spmasks = self._batch()
            self.optimizer.zero_grad()
            preds = self.forward_train(images)
            group_loss = self.group_multi_loss(preds, labels, superpixels, spmasks)
            pos_loss = self.multi_pos_loss(preds, labels, superpixels, spmasks)
            loss = self.args.coeff * pos_loss + group_loss
            self.update(loss)
            self.update_average_meter({'train-loss': loss, 'pos-loss': pos_loss, 'group-loss': group_loss})
            self.log_training(iteration, None, total_itrs)
            self.log_validation(iteration, val_period)
        self.flush_meters()
        self.check_stream_k()
