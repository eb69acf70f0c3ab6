"""Model factory with the reference's surface -- ``models/__init__.py:21-51``:
``get_model(model, num_classes, output_stride, separable_conv)`` returns a module with ``.backbone``,
``.classifier``, ``.forward``, ``.feat_forward``, ``.set_return_feat`` whose ``state_dict`` key names
equal the reference's, so the authors' checkpoints load.  In scope: the production architecture
``deeplabv3pluswn_resnet50deepstem`` (and its ResNet-101 twin, same code)."""
import torch
import torch.nn as nn

from .deeplab import DeepLabV3PlusWN, build_deeplabv3pluswn  # noqa: F401  (DeepLabV3PlusWN re-exported)

_ARCHS = {
    'deeplabv3pluswn_resnet50deepstem': ((3, 4, 6, 3), './checkpoint/resnet50_deepstem.pth'),
    'deeplabv3pluswn_resnet101deepstem': ((3, 4, 23, 3), './checkpoint/resnet101_deepstem.pth'),
}


def set_bn_momentum(model, momentum=0.1):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = momentum


def freeze_bn(model):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.requires_grad_(False)
            m.bias.requires_grad_(False)
            m.eval()


def get_model(model, num_classes, output_stride, separable_conv, pretrained_backbone=True):
    """``pretrained_backbone=True`` reproduces the reference: the ImageNet deep-stem backbone is read from
    ``./checkpoint/resnet50_deepstem.pth`` (``backbone/resnet.py:306``; FileNotFoundError if absent)."""
    if model not in _ARCHS:
        raise NotImplementedError("only %s are on the hot path (SURVEY.md section 2.1 #5), got %r" % (sorted(_ARCHS), model))
    layers, ckpt = _ARCHS[model]
    net = build_deeplabv3pluswn(num_classes, output_stride, layers, separable=bool(separable_conv))
    if pretrained_backbone:
        state = torch.load(ckpt, map_location='cpu')
        if isinstance(state, dict) and 'model' in state:
            state = state['model']
        net.backbone.load_state_dict(state, strict=False)
    set_bn_momentum(net.backbone, momentum=0.1)
    return net
