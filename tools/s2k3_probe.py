"""3x3 stride-2 forward of a training step: the stride-2 form of k_conv_bx (nine shifted 1x1 products) against the stream-K f32 kernel
(with its BatchNorm statistics epilogue) and the BatchNorm reduction pass the split-bf16 form leaves to bn_act_train."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mulactseg_amd import ops
from conv_table import timeit

dev = torch.device('cuda:0')
for (N, C, H, W, M) in ((4, 128, 192, 192, 128), (4, 256, 96, 96, 256), (4, 128, 256, 512, 128), (4, 256, 128, 256, 256)):
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(M, C, 3, 3, device=dev) * 0.05
    pk = ops.conv_bx_pack(w, ops.BX_ROLE_S2)
    psk = ops.packed_weight(w, 2, False)
    t_bx = timeit(lambda: ops.conv_bx_s2_raw(x, w, packed=pk))
    t_sk = timeit(lambda: ops.conv_sk(x, w, 2, 1, packed=psk))
    t_sks = timeit(lambda: ops.conv_sk(x, w, 2, 1, packed=psk, stats=True))
    y = ops.conv_bx_s2_raw(x, w, packed=pk)
    bn = torch.nn.BatchNorm2d(M).to(dev).train()
    t_bn = timeit(lambda: ops.bn_act(bn, y, relu=True))
    _, part = ops.conv_sk(x, w, 2, 1, packed=psk, stats=True)
    t_bns = timeit(lambda: ops.bn_act(bn, y, relu=True, partials=part))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 2, 1).float()
    print((N, C, H, W, M), "bx %.1f us  sk %.1f us  sk+stats %.1f us | bn with own reduction %.1f us, with partials %.1f us | err %.2e"
          % (t_bx, t_sk, t_sks, t_bn, t_bns, float((y - ref).abs().max() / ref.abs().max())))
