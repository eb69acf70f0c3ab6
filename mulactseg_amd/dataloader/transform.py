"""Training transforms by name -- the reference's ``dataloader/transform.py:5-170`` (Cityscapes) and
``dataloader/transform_voc.py:5-224`` (VOC) -- built on the device augmentation (``device_transforms.py``,
``csrc/augment.hip``): the names the launch scripts use (``script/**/*.sh``: ``rescale_769_multi_notrg``, ``rescale_769_nospx``,
``rescale_513_multi_notrg``, ``rescale_513_notrg``, ``eval_spx``) plus the other names with the same structure.  A transform is
called as ``transform(picture u8 [H,W,3] cuda, [maps]) -> (image f32 [3,h,w], [maps])``; ``transform.n_maps`` says how many maps
it pads (``ExtRandomCrop.pad_values``; the reference asserts the same count, ``ext_transforms.py:489``).

Not offered (outside the production configurations): the unpadded 512x1024 crops (``orig_*``, ``rescale``), ``load_smaller_spx``
(a third map), the colour-jitter variant, the multi-scale identity evaluation (``eval_spx_identity_ms``)."""
from .device_transforms import DeviceResize, DeviceResizeFlip, DeviceTrainAugment


def _with_maps(t, n):
    t.n_maps = n
    return t


def _no_small(args, name):
    if getattr(args, 'load_smaller_spx', False):
        raise NotImplementedError("train_transform %r with --load_smaller_spx (a third map) is outside the hot path" % name)


def get_train_transform(args, transform):
    """Cityscapes: 768x768 crops of a U(0.5, 2) rescale, padded with (124, 116, 104) / the per-map pad values."""
    if transform is None:
        return None
    crop = dict(size=(768, 768), scale_range=(0.5, 2.0))
    if transform == 'rescale_769_nospx':                            # [label]
        return _with_maps(DeviceTrainAugment(pad_values=[args.ignore_idx], **crop), 1)
    if transform == 'rescale_769':                                  # [label, superpixel]
        return _with_maps(DeviceTrainAugment(pad_values=[args.ignore_idx, args.nseg], **crop), 2)
    if transform == 'rescale_769_multi':
        _no_small(args, transform)
        return _with_maps(DeviceTrainAugment(pad_values=[args.ignore_idx, args.nseg], **crop), 2)
    if transform == 'rescale_769_multi_notrg':                      # [superpixel]: the stage-1 production transform
        _no_small(args, transform)
        return _with_maps(DeviceTrainAugment(pad_values=[args.nseg], **crop), 1)
    if transform == 'rescale_769_multi_notrg_ignore':               # [label padded with 0, superpixel]
        _no_small(args, transform)
        return _with_maps(DeviceTrainAugment(pad_values=[0, args.nseg], **crop), 2)
    if transform in ('eval_spx', 'eval_dom_gt_spx'):                # ExtResize((1024, 2048)), two maps
        return _with_maps(DeviceResize((1024, 2048), pad_values=[args.ignore_idx, args.nseg]), 2)
    raise NotImplementedError("train_transform %r is outside the hot path (see dataloader/transform.py)" % transform)


def get_train_transform_voc(args, transform):
    """VOC: 513x513 crops."""
    if transform is None:
        return None
    crop = dict(size=(513, 513), scale_range=(0.5, 2.0))
    if transform == 'rescale_769_nospx':                            # resize 513 + centre crop 513 + flip, [label]
        return _with_maps(DeviceResizeFlip(513, center_crop=513, pad_values=[args.ignore_idx]), 1)
    if transform == 'rescale_513_notrg':                            # [label]
        return _with_maps(DeviceTrainAugment(pad_values=[args.ignore_idx], **crop), 1)
    if transform == 'rescale_513':                                  # [label, superpixel]
        return _with_maps(DeviceTrainAugment(pad_values=[args.ignore_idx, args.nseg], **crop), 2)
    if transform == 'rescale_513_multi_notrg':                      # [superpixel]
        _no_small(args, transform)
        return _with_maps(DeviceTrainAugment(pad_values=[args.nseg], **crop), 1)
    if transform == 'eval_spx':
        return _with_maps(DeviceResize(513, center_crop=513, pad_values=[args.ignore_idx, args.nseg]), 2)
    raise NotImplementedError("train_transform %r is outside the hot path (see dataloader/transform.py)" % transform)


def get_val_transform(name, n_maps=1, ignore_idx=255, nseg=2048):
    """The pool / validation / evaluation transform of ``dataloader/__init__.py:38-78,124-136,156-170``."""
    pads = [ignore_idx, nseg]            # (never applied: a resize / centre crop pads nothing)
    if name == 'cityscapes':
        return _with_maps(DeviceResize((1024, 2048), pad_values=pads), n_maps)
    if name == 'voc':
        return _with_maps(DeviceResize(513, center_crop=513, pad_values=pads), n_maps)
    raise NotImplementedError(name)
