"""DeepLabv3+ with the weight-normalised (cosine) classifier over a deep-stem ResNet, OS 16.

Architecture and parameter names follow the reference so that checkpoints interchange:
  backbone   ``models/segmentation/backbone/resnet.py:119-232`` (deep stem 3x(3x3), Bottleneck x [3,4,6,3],
             layer4 dilated for output stride 16, ``modeling.py:11-16``), wrapped like
             ``IntermediateLayerGetter`` (``utils.py:45-100``): returns layer1 ('low_level') and layer4 ('out');
  head       ``DeepLabHeadV3PlusWN`` (``deeplabv3.py:85-137``): low-level 1x1 -> 48, ASPP (1x1 | three atrous
             separable 3x3, d = 6/12/18 | image pooling) -> 1x1 256 + Dropout(0.1), bilinear x4, concat 304 ->
             two atrous-separable 3x3 256, then cosine similarity with the class proxies;
  wrapper    ``_SimpleSegmentationModel`` (``utils.py:6-42``): bilinear upsample to the input size.
``convert_to_separable_conv`` (``deeplabv3.py:249-261``) semantics are built in: every k > 1 conv of the HEAD
is depthwise(k, dilation, no bias) followed by pointwise 1x1 (no bias), with nothing in between.

On the GPU the dense convolutions run on this package's matrix-core kernels.  Wherever the shape allows, the f32 products are computed
on the bf16 matrix cores from exact three-term splits of both operands (csrc/bx_split.h): at inference csrc/conv_bx.hip with the
BatchNorm, residual add and ReLU in its epilogue (csrc/conv_mfma.hip, f32 MFMA, for the two stride-2 3x3 layers; no MIOpen kernel in a
pool forward); in training csrc/conv_bx.hip for the forward product and the input gradient of every stride-1 layer (K chunks of the
small planes' layers dealt to several workgroups), csrc/conv_wgrad_bx.hip / conv_wgrad_bx3.hip for the 1x1 / 3x3 weight gradients,
and the f32-MFMA kernels csrc/conv_sk.hip (persistent stream-K) / csrc/conv_wgrad.hip for the stride-2 and the narrow layers, as
ops.conv_train_plan / ops.conv_bx_train_ok select them (path_report() says which products of which layer took which kernel).  The memory-bound layers around them take the HIP
kernels of this package too: BatchNorm + ReLU + residual add (csrc/bn.hip), every depthwise 3x3 (csrc/aspp.hip; the three ASPP
dilations from one read of the feature map), the bilinear upsamplings (csrc/upsample.hip).  On the CPU the same modules run as
plain PyTorch ops (parity tests against the executed reference, tests/test_model.py).
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# Which implementation every fusable layer call took, per process: {("bn_act", "hip"): n, ("bn_act", "aten"): m, ...}.
# On the GPU a layer falls back to the ATen / MIOpen op only when ops.*_supported() declines its shape or mode (e.g. a
# frozen BatchNorm with gradients flowing through it); bench.py prints the table so a run shows what it measured.
PATH_COUNTS = {}


def _took(kind, path):
    PATH_COUNTS[(kind, path)] = PATH_COUNTS.get((kind, path), 0) + 1


def path_report(reset=False):
    """{"bn_act": {"hip": n, "aten": m}, ...} since the last reset."""
    out = {}
    for (kind, path), n in sorted(PATH_COUNTS.items()):
        out.setdefault(kind, {})[path] = n
    if reset:
        PATH_COUNTS.clear()
    return out


def _bn_act(bn, x, relu=True, residual=None):
    """relu(bn(x) + residual): on the GPU one fused pass (csrc/bn.hip) instead of BatchNorm, add and ReLU kernels."""
    if x.is_cuda:
        from .. import ops
        if ops.bn_act_supported(bn, x, residual):
            _took("bn_act", "hip")
            return ops.bn_act(bn, x, relu, residual)
    _took("bn_act", "aten")
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def _INFER_BX():
    """MAS_INFER_CONV=f32 keeps every inference convolution on the f32 matrix cores (csrc/conv_mfma.hip) for A/B runs."""
    return os.environ.get("MAS_INFER_CONV", "bx") != "f32"


def _conv_bn_act(conv, bn, x, relu=True, residual=None, fork=False):
    """relu?(bn(conv(x)) + residual); fork=True: (that, x') with x' = x for the OTHER consumer of x (in training on the package's
    kernels an alias of x through which that consumer's gradient reaches the epilogue of this convolution's input-gradient kernel).  Inference on the GPU: ONE kernel on the f32 matrix cores with the BatchNorm, the
    residual add and the ReLU in its epilogue (csrc/conv_mfma.hip) -- no separate normalisation pass over the activation.
    Training (batch statistics, autograd) and unsupported geometries: MIOpen's convolution + the fused BatchNorm kernels."""
    want_fork, x_other = fork, x
    if x.is_cuda and not bn.training and bn.track_running_stats and not torch.is_grad_enabled():
        if fork:
            return _conv_bn_act(conv, bn, x, relu, residual), x
        from .. import ops
        if _INFER_BX() and x.shape[2] * x.shape[3] >= 256 and ops.conv_bx_supported(conv, x):
            # bf16 matrix cores, f32 operands split exactly into three terms (csrc/conv_bx.hip): f32 error bound, 1.5-2x the f32 MFMA kernel
            _took("conv_bn_act", "hip_bx")
            return ops.conv_bx(conv, x, bn, relu, residual)
        if x.shape[2] * x.shape[3] >= 256 and ops.conv_mfma_supported(conv, x):
            _took("conv_bn_act", "hip_mfma")
            return ops.conv_mfma(conv, x, bn, relu, residual)
        if residual is None and ops.stem_conv_supported(conv, x):          # 3 -> 64, stride 2: output-bound, packed-f32 kernel
            _took("conv_bn_act", "hip_stem")
            return ops.stem_conv(conv, x, bn, relu)
        if (residual is None and x.shape[2] * x.shape[3] == 1 and conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
                and conv.bias is None and x.dtype == torch.float32):
            # a 1x1 convolution of a 1x1 map (the ASPP image-pooling branch) is a [N,K] x [K,M] product: the fixed-order kernel of
            # csrc/head.hip (no vendor GEMM, no MIOpen solver search for every new batch size)
            _took("conv_bn_act", "dense1x1")
            scale, shift = ops._bn_fold(bn)
            y = ops.conv1x1_on_1x1(conv, x)[:, :, 0, 0] * scale + shift
            return (F.relu(y) if relu else y)[:, :, None, None]
    if x.is_cuda and torch.is_grad_enabled():
        from .. import ops
        if os.environ.get("MAS_TRAIN_CONV", "own") != "miopen" and ops.conv1x1_on_1x1_supported(conv, x):
            # the ASPP image-pooling branch: 2 MFLOP; a fixed-order product (the vendor GEMM / MIOpen solvers for this shape use
            # atomic split-K: run-to-run different bits, amplified by the BatchNorm over N samples behind it)
            _took("conv_bn_act", "train:dense1x1")
            out = _bn_act(bn, ops.conv1x1_on_1x1(conv, x), relu, residual)
            return (out, x_other) if want_fork else out
        own = ops.conv_train_plan(conv, x)          # training: (forward, input gradient, weight gradient) on the f32-MFMA kernels
        if own is not None and any(own):
            # "/bx": forward (and with it the input gradient) on the split-bf16 kernel csrc/conv_bx.hip instead of the stream-K one
            bx = own[0] and ops.conv_bx_train_ok(x.shape, conv.weight.shape, conv.stride[0], conv.dilation[0], False)
            _took("conv_bn_act", "train:" + "".join(n if o else "-" for n, o in zip("fdw", own)) + ("/bx" if bx else ""))
            # stats: the forward kernel forms the BatchNorm partial sums of its output in its epilogue (no reduction pass over y);
            # fork: the gradient of x's other consumer is added in the epilogue of this convolution's input-gradient kernel
            # (round 5: the split-bf16 kernel forms them too, unless its plan splits K -- then `part` comes back None and the BatchNorm
            #  runs its own reduction pass)
            stats = own[0] and bn.training and os.environ.get("MAS_BN_STATS", "fused") == "fused" and not (bx and os.environ.get("MAS_BX_STATS", "on") == "off")
            fork = fork and own[1] and conv.stride[0] == 1 and x.requires_grad and os.environ.get("MAS_GRAD_FORK", "fused") == "fused"
            res = ops.conv_train(conv, x, own, stats=stats, fork=fork)
            y, part = (res[0], res[1]) if (stats or fork) else (res, None)
            if fork:
                x_other = res[2]
            if part is not None and ops.bn_act_supported(bn, y, residual):
                _took("bn_act", "hip")
                out = ops.bn_act(bn, y, relu, residual, partials=part)
            else:
                out = _bn_act(bn, y, relu, residual)
            return (out, x_other) if want_fork else out
    _took("conv_bn_act", "miopen+bn")
    out = _bn_act(bn, conv(x), relu, residual)
    return (out, x_other) if want_fork else out


def _run(seq, x):
    """nn.Sequential forward with every (Conv2d, BatchNorm2d[, ReLU]) triple and (BatchNorm2d, ReLU) pair fused; other
    modules run as they are."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            x = _conv_bn_act(m, mods[i + 1], x, relu)
            i += 3 if relu else 2
        elif hasattr(m, 'depthwise') and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)           # separable: depthwise, then pointwise + BN (+ ReLU) fused
            x = _conv_bn_act(m.body[1], mods[i + 1], m.depthwise(x), relu)
            i += 3 if relu else 2
        elif isinstance(m, nn.BatchNorm2d):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = _bn_act(m, x, relu)
            i += 2 if relu else 1
        else:
            x = m(x)
            i += 1
    return x


# ------------------------------------------------------------------------------------------------
# backbone
# ------------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = None
        if x.is_cuda and not self.training:
            from .. import ops
            if _INFER_BX() and x.shape[2] * x.shape[3] >= 256 and ops.conv_bx_supported(self.conv1, x):
                pass                                                      # (the split-bf16 kernel is faster than the VALU one below)
            elif ops.conv1x1_bn_act_supported(self.conv1, self.bn1, x):     # small-K 1x1 + BN + ReLU in one kernel (csrc/conv1x1.hip)
                _took("conv1x1_bn_act", "hip")
                y = ops.conv1x1_bn_act(self.conv1, self.bn1, x, True)
        if y is None:
            # x has two consumers (conv1 and the residual branch): see _conv_bn_act(fork=True)
            y, x = _conv_bn_act(self.conv1, self.bn1, x, fork=True)
        if self.downsample is not None and x.is_cuda and self.training and torch.is_grad_enabled():
            from .. import ops
            st = ops.branch_stream(x.device, 0)
            if st is not None:
                # training: the identity branch on its own stream beside conv2 (forward) / beside the conv3-conv2 chain (backward)
                idn = ops.run_on_branch(st, lambda t: _run(self.downsample, t), x)
                y = _conv_bn_act(self.conv2, self.bn2, y)
                ops.join_branch(st, idn)
                return _conv_bn_act(self.conv3, self.bn3, y, True, idn)
        y = _conv_bn_act(self.conv2, self.bn2, y)
        if self.downsample is not None:
            if (x.is_cuda and not self.training and not torch.is_grad_enabled() and _INFER_BX() and len(self.downsample) == 2
                    and os.environ.get("MAS_INFER_DUAL", "on") != "off"):
                from .. import ops
                ds_conv, ds_bn = self.downsample[0], self.downsample[1]
                if (isinstance(ds_conv, nn.Conv2d) and isinstance(ds_bn, nn.BatchNorm2d) and not ds_bn.training and not self.bn3.training
                        and ops.conv_bx_dual_supported(self.conv3, y, ds_conv, x)):
                    # stride-1 downsample (layer1.0; layer4.0 at output stride 16): conv3 + bn3 and downsample + bn in one kernel, one
                    # accumulator set -- the identity branch is never written and read back
                    _took("conv_bn_act", "hip_bx_dual")
                    return ops.conv_bx_dual(self.conv3, self.bn3, y, ds_conv, ds_bn, x, True)
            x = _run(self.downsample, x)
        return _conv_bn_act(self.conv3, self.bn3, y, True, x)


class DeepStemResNetTrunk(nn.Module):
    """conv1 (3 convs) / bn1 / relu / maxpool / layer1..layer4; returns {'low_level', 'out'}."""

    def __init__(self, layers=(3, 4, 6, 3), stem_width=64, output_stride=16):
        super().__init__()
        if output_stride == 16:
            dilate = (False, False, True)
        elif output_stride == 8:
            dilate = (False, True, True)
        else:
            raise ValueError("output_stride must be 8 or 16")
        w = stem_width
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, w, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(w), nn.ReLU(inplace=True),
            nn.Conv2d(w, w, 3, stride=1, padding=1, bias=False), nn.BatchNorm2d(w), nn.ReLU(inplace=True),
            nn.Conv2d(w, 2 * w, 3, stride=1, padding=1, bias=False))
        self.bn1 = nn.BatchNorm2d(2 * w)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self._inplanes, self._dilation = 2 * w, 1
        self.layer1 = self._stage(64, layers[0], 1, False)
        self.layer2 = self._stage(128, layers[1], 2, dilate[0])
        self.layer3 = self._stage(256, layers[2], 2, dilate[1])
        self.layer4 = self._stage(512, layers[3], 2, dilate[2])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, planes, blocks, stride, dilate):
        prev = self._dilation
        if dilate:
            self._dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self._inplanes != planes * Bottleneck.expansion:
            down = nn.Sequential(nn.Conv2d(self._inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * Bottleneck.expansion))
        mods = [Bottleneck(self._inplanes, planes, stride, prev, down)]
        self._inplanes = planes * Bottleneck.expansion
        mods += [Bottleneck(self._inplanes, planes, 1, self._dilation) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, x):
        stem = list(self.conv1)
        x = _conv_bn_act(stem[-1], self.bn1, _run(stem[:-1], x))
        if x.is_cuda:
            from .. import ops
            hip = ops.maxpool3s2_supported(self.maxpool, x)
            _took("maxpool", "hip" if hip else "aten")
            x = ops.maxpool3s2(x) if hip else self.maxpool(x)
        else:
            x = self.maxpool(x)
        low = self.layer1(x)
        out = self.layer4(self.layer3(self.layer2(low)))
        return OrderedDict(low_level=low, out=out)


# ------------------------------------------------------------------------------------------------
# head
# ------------------------------------------------------------------------------------------------
class AtrousSeparableConvolution(nn.Module):
    """depthwise kxk (dilated) -> pointwise 1x1, both bias-free, nothing in between
    (``deeplabv3.py:168-192`` as instantiated by ``convert_to_separable_conv``)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding, dilation):
        super().__init__()
        self.body = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, kernel_size, padding=padding, dilation=dilation, bias=False, groups=in_channels),
            nn.Conv2d(in_channels, out_channels, 1, bias=False))

    def depthwise(self, x):
        dw = self.body[0]
        if x.is_cuda and dw.kernel_size == (3, 3) and dw.stride == (1, 1) and dw.padding == dw.dilation and dw.groups == x.shape[1]:
            from .. import ops          # HIP depthwise (csrc/aspp.hip); MIOpen only has a naive fp32 kernel for it
            if ops.depthwise3x3_supported(x, dw.dilation[0]):
                _took("depthwise3x3", "hip")
                return ops.depthwise3x3(x, dw.weight, dw.dilation[0])
        _took("depthwise3x3", "miopen")
        return dw(x)

    def forward(self, x):
        return self.body[1](self.depthwise(x))


def _conv3x3(cin, cout, dilation, separable):
    if separable:
        return AtrousSeparableConvolution(cin, cout, 3, padding=dilation, dilation=dilation)
    return nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation, bias=False)


def _upsample(x, size):
    """F.interpolate(bilinear, align_corners=False); on the GPU the HIP kernels of csrc/upsample.hip (same index and
    weight arithmetic, deterministic gather backward instead of ATen's atomic scatter)."""
    if x.is_cuda:
        from .. import ops
        if ops.upsample_bilinear_supported(x, size):
            _took("upsample", "hip")
            return ops.upsample_bilinear(x, size)
    _took("upsample", "aten")
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


class _ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        # bilinear upsampling of a 1x1 map is a broadcast (F.interpolate differs from it by the rounding of l0*v + l1*v: 2e-7);
        # ATen's backward for it is 2 304 atomic adds into ONE element per channel: 1.1 ms per step.  On the CPU the reference's
        # own op (deeplabv3.py:206-207), so that the CPU form of the network equals the executed reference bit for bit (G10).
        y = _run(self, x)
        if not x.is_cuda:
            return F.interpolate(y, size=x.shape[-2:], mode='bilinear', align_corners=False)
        return y.expand(-1, -1, x.shape[-2], x.shape[-1])


class ASPP(nn.Module):
    def __init__(self, cin, rates, separable):
        super().__init__()
        cout = 256
        branches = [nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))]
        for r in rates:
            branches.append(nn.Sequential(_conv3x3(cin, cout, r, separable), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)))
        branches.append(_ASPPPooling(cin, cout))
        self.convs = nn.ModuleList(branches)
        self.project = nn.Sequential(nn.Conv2d(5 * cout, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                                     nn.Dropout(0.1))

    def _fused_depthwise(self, x):
        """K7: the three dilated depthwise convolutions from ONE read of x (csrc/aspp.hip) when the branches are
        separable and x lives on the GPU; parameters and state-dict layout are untouched."""
        seps = [self.convs[i][0] for i in (1, 2, 3)]
        if not (x.is_cuda and x.dtype == torch.float32 and all(isinstance(m, AtrousSeparableConvolution) for m in seps)):
            return None
        dws = [m.body[0] for m in seps]
        if any(d.kernel_size != (3, 3) or d.stride != (1, 1) or d.padding != d.dilation or d.groups != x.shape[1] for d in dws):
            return None
        from .. import ops
        _took("aspp_depthwise_triple", "hip")
        return ops.aspp_depthwise3(x, dws[0].weight, dws[1].weight, dws[2].weight, [d.dilation[0] for d in dws])

    def forward(self, x):
        fused = self._fused_depthwise(x)
        if fused is None:                          # CPU reference form (parity tests), or a non-separable head
            return self.project(torch.cat([conv(x) for conv in self.convs], dim=1))
        streams = [None] * 5
        if x.is_cuda and self.training and torch.is_grad_enabled():
            from .. import ops
            # training: the five branches are independent 2048 -> 256 products on 48 x 48 planes (144 workgroups each): three streams
            streams = [None, ops.branch_stream(x.device, 1), ops.branch_stream(x.device, 2), None, ops.branch_stream(x.device, 1)]

        def on(k, fn, t):
            if streams[k] is None:
                return fn(t)
            from .. import ops
            return ops.run_on_branch(streams[k], fn, t)
        outs = [None] * 5
        for i, y in zip((1, 2, 3), fused):
            branch = self.convs[i]
            outs[i] = on(i, lambda t, b=branch: _conv_bn_act(b[0].body[1], b[1], t), y)      # pointwise 1x1 -> BN + ReLU
        outs[4] = on(4, self.convs[4], x)
        outs[0] = on(0, lambda t: _run(self.convs[0], t), x)
        for k in (1, 2, 4):
            if streams[k] is not None:
                from .. import ops
                ops.join_branch(streams[k], outs[k])
        return _run(self.project, torch.cat(outs, dim=1))


class DeepLabHeadV3PlusWN(nn.Module):
    def __init__(self, in_channels, low_level_channels, num_classes, aspp_dilate, separable):
        super().__init__()
        self.project = nn.Sequential(nn.Conv2d(low_level_channels, 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU(inplace=True))
        self.aspp = ASPP(in_channels, aspp_dilate, separable)
        self.classifier = nn.Sequential(
            _conv3x3(304, 256, 1, separable), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            _conv3x3(256, 256, 1, separable), nn.BatchNorm2d(256), nn.ReLU(inplace=True))
        self.final = nn.Conv2d(256, num_classes, 1, bias=False)
        self.proxy = self.final.weight          # the same Parameter under two names (deeplabv3.py:88-89)
        self.return_feat = False
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def point_feature(self, feature):
        st = None
        if feature['low_level'].is_cuda and self.training and torch.is_grad_enabled():
            from .. import ops
            st = ops.branch_stream(feature['low_level'].device, 3)
        if st is not None:          # training: the low-level projection (256 -> 48 at 192 x 192) beside the ASPP
            low = ops.run_on_branch(st, lambda t: _run(self.project, t), feature['low_level'])
            x = self.aspp(feature['out'])
            ops.join_branch(st, low)
        else:
            low = _run(self.project, feature['low_level'])
            x = self.aspp(feature['out'])
        if x.is_cuda and not torch.is_grad_enabled():
            from .. import ops
            if ops.upsample_bilinear_supported(x, low.shape[2:]):
                # inference: the upsampled ASPP output is written straight into its slice of the concatenation buffer
                # (torch.cat re-reads and re-writes both inputs: 0.4 ms per pool batch)
                _took("upsample", "hip")
                N, cl = low.shape[0], low.shape[1]
                buf = torch.empty((N, cl + x.shape[1], low.shape[2], low.shape[3]), dtype=low.dtype, device=low.device)
                buf[:, :cl].copy_(low)
                ops.upsample_bilinear_into(x.contiguous(), buf[:, cl:])
                return _run(self.classifier, buf)
        x = _upsample(x, low.shape[2:])
        return _run(self.classifier, torch.cat([low, x], dim=1))

    def forward(self, feature, return_feat=None):
        """``return_feat``: None = the module's flag (the reference's ``set_return_feat`` protocol); True / False = for this call only
        (callers that run on several threads must not flip a flag on the shared module)."""
        want_feat = self.return_feat if return_feat is None else return_feat
        pf = self.point_feature(feature)
        if pf.is_cuda:
            from .. import ops
            if ops.cosine_head_supported(pf, self.proxy):
                _took("cosine_head", "hip")
                out = ops.cosine_head(pf, self.proxy)            # K8: one pass over the features (csrc/head.hip)
                return (F.normalize(pf), out) if want_feat else out
        feat = F.normalize(pf)                                   # over channels, eps 1e-12
        out = F.conv2d(feat, F.normalize(self.proxy, dim=1))     # cosine similarity in [-1, 1]
        return (feat, out) if want_feat else out


class DeepLabV3PlusWN(nn.Module):
    lowres_logits = True        # forward(x, lowres=True) returns the quarter-resolution logits (selectors / losses upsample in-kernel)

    def __init__(self, backbone, classifier):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier
        self.return_feat = False

    def set_return_feat(self):
        self.return_feat = True
        self.classifier.return_feat = True

    def unset_return_feat(self):
        self.return_feat = False
        self.classifier.return_feat = False

    def quarter_logits(self, x):
        """Cosine logits at 1/4 resolution (before the final bilinear upsample)."""
        keep = self.classifier.return_feat
        self.classifier.return_feat = False
        try:
            return self.classifier(self.backbone(x))
        finally:
            self.classifier.return_feat = keep

    def forward(self, x, lowres=False):
        """Logits at the input size (``utils.py:25``).  ``lowres=True`` returns the quarter-resolution cosine logits the final
        bilinear upsampling starts from -- for consumers that evaluate it themselves (``FusedPartialLabelLoss.forward_lowres``)."""
        size = x.shape[-2:]
        y = self.classifier(self.backbone(x))
        if lowres:
            return y[1] if isinstance(y, tuple) else y
        return _upsample(y, size)

    def feat_forward_lowres(self, x):
        """(L2-normalised point features at 1/4 resolution [N,256,H/4,W/4], logits upsampled to the input size):
        what the stage-2 pseudo-label kernels consume -- they interpolate the features per pixel instead of
        materialising feat_forward's 256-channel full-resolution tensor (2.1 GB per Cityscapes image)."""
        size = x.shape[-2:]
        feat, prob = self.classifier(self.backbone(x), return_feat=True)
        return feat, _upsample(prob, size)

    def feat_forward(self, x):
        size = x.shape[-2:]
        feat, prob = self.classifier(self.backbone(x), return_feat=True)
        return _upsample(feat, size), _upsample(prob, size)


def build_deeplabv3pluswn(num_classes, output_stride=16, layers=(3, 4, 6, 3), separable=True):
    rates = (12, 24, 36) if output_stride == 8 else (6, 12, 18)
    backbone = DeepStemResNetTrunk(layers, 64, output_stride)
    head = DeepLabHeadV3PlusWN(2048, 256, num_classes, rates, separable)
    return DeepLabV3PlusWN(backbone, head)
