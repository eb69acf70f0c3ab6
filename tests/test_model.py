"""Model parity (row a-10): our DeepLabv3+WN / ResNet50-deepstem against the executed reference
(tests/golden/g4_model.npz): identical state-dict key names and shapes, cosine logits within 1e-4
(north-star tolerance) in eval mode -- on CPU here and on the GPU (MIOpen) under -m gpu."""
import hashlib
import os

import numpy as np
import pytest
import torch

from mulactseg_amd import synth
from mulactseg_amd.models import get_model

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    g = np.load(os.path.join(GOLDEN, "g4_model.npz"))
    net = get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True, pretrained_backbone=False)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert sorted(shapes) == list(g['keys'])                                   # reference checkpoints load
    assert [str(shapes[k]) for k in sorted(shapes)] == list(g['shapes'])
    sd = synth.synthetic_state_dict(shapes, seed=int(g['seed']))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.eval()
    x = np.random.RandomState(int(g['x_seed'])).standard_normal(size=(1, 3, 129, 161)).astype(np.float32)
    h = hashlib.sha256(); h.update(x.tobytes())
    assert np.frombuffer(h.digest()[:8], dtype=np.uint64)[0] == g['input_digest']
    return g, net, torch.from_numpy(x)


def _check(g, net, x, tol):
    with torch.no_grad():
        feats = net.backbone(x)
        quarter = net.quarter_logits(x)
        full = net(x)
        net.set_return_feat()
        feat_up, prob_up = net.feat_forward(x)
        net.unset_return_feat()
    for name, ref in (('low_level', float(g['low_level_mean'])), ('out', float(g['out_mean']))):
        assert abs(float(feats[name].double().mean()) - ref) < 10 * tol * max(1.0, abs(ref))   # un-normalised features
    assert np.max(np.abs(quarter.cpu().numpy() - g['quarter'])) < tol
    assert np.max(np.abs(full[:, :, ::3, ::3].cpu().numpy() - g['full_sub'])) < tol
    assert np.max(np.abs(prob_up[:, :, ::3, ::3].cpu().numpy() - g['prob_up_sub'])) < tol
    assert np.max(np.abs(feat_up[:, ::16, ::5, ::5].cpu().numpy() - g['feat_up_sub'])) < tol
    assert float(quarter.abs().max()) <= 1.0 + 1e-5                              # cosine similarity


def test_model_matches_reference_cpu():
    g, net, x = _load()
    _check(g, net, x, 1e-5)
    assert net.classifier.proxy is net.classifier.final.weight                  # one Parameter, two names
    assert sum(p.numel() for p in net.parameters()) == 26806224


@pytest.mark.gpu
def test_model_matches_reference_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    g, net, x = _load()
    _check(g, net.cuda(), x.cuda(), 1e-4)


def test_get_model_surface():
    with pytest.raises(FileNotFoundError):
        get_model('deeplabv3pluswn_resnet50deepstem', 20, 16, True)            # reference: ./checkpoint/resnet50_deepstem.pth
    with pytest.raises(NotImplementedError):
        get_model('deeplabv3_mobilenet', 20, 16, True)


@pytest.mark.gpu
def test_inference_forward_runs_on_this_packages_kernels_and_matches_the_cpu_modules():
    """Aligned picture size (the Cityscapes / crop sizes): every dense convolution of the eval forward takes a kernel of this
    package (k_conv_mfma with the BatchNorm epilogue, k_stem_conv, k_conv1x1, a GEMM for the pooled 1x1) -- no MIOpen
    convolution, no separate BatchNorm pass -- and the logits equal the plain PyTorch modules on the CPU to 1e-4."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.models import deeplab
    g, net, _ = _load()
    x = torch.from_numpy(np.random.RandomState(5).standard_normal(size=(1, 3, 256, 512)).astype(np.float32))
    with torch.no_grad():
        ref_full = net(x)
        ref_q = net(x, lowres=True)
        dev = net.cuda()
        deeplab.path_report(reset=True)
        out_full = dev(x.cuda())
        paths = deeplab.path_report(reset=True)
        out_q = dev(x.cuda(), lowres=True)
    assert 'miopen+bn' not in paths.get('conv_bn_act', {}), paths
    took = paths['conv_bn_act']
    assert took.get('hip_mfma', 0) + took.get('hip_bx', 0) >= 55 and took.get('hip_stem', 0) == 1, paths
    assert took.get('hip_bx', 0) >= 40, paths           # the split-bf16 kernel (csrc/conv_bx.hip) takes every shape it supports
    assert took.get('hip_bx_dual', 0) == 2, paths       # conv3 + stride-1 downsample of layer1.0 / layer4.0 as one kernel
    assert 'aten' not in paths.get('bn_act', {}) and 'aten' not in paths.get('upsample', {})
    assert float((out_full.cpu() - ref_full).abs().max()) < 1e-4
    assert float((out_q.cpu() - ref_q).abs().max()) < 1e-4


@pytest.mark.gpu
def test_split_bf16_and_f32_matrix_core_forwards_agree(monkeypatch):
    """The same eval forward with every dense convolution on the f32 matrix cores (MAS_INFER_CONV=f32, csrc/conv_mfma.hip) and with
    the split-bf16 kernel (csrc/conv_bx.hip) where it applies: both within 1e-4 of the CPU modules, and the split form is not
    further from them than the f32 form by more than rounding noise."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.models import deeplab
    g, net, _ = _load()
    x = torch.from_numpy(np.random.RandomState(7).standard_normal(size=(2, 3, 256, 512)).astype(np.float32))
    with torch.no_grad():
        ref = net(x, lowres=True)
        dev = net.cuda()
        monkeypatch.setenv("MAS_INFER_CONV", "f32")
        deeplab.path_report(reset=True)
        out32 = dev(x.cuda(), lowres=True).cpu()
        assert 'hip_bx' not in deeplab.path_report(reset=True)['conv_bn_act']
        monkeypatch.setenv("MAS_INFER_CONV", "bx")
        outbx = dev(x.cuda(), lowres=True).cpu()
        assert deeplab.path_report(reset=True)['conv_bn_act'].get('hip_bx', 0) >= 40
    e32, ebx = float((out32 - ref).abs().max()), float((outbx - ref).abs().max())
    assert e32 < 1e-4 and ebx < 1e-4, (e32, ebx)
    assert ebx <= 2.0 * e32 + 2e-6, (e32, ebx)
