"""Developer probe: mas_conv1x1_fwd against F.conv2d (MIOpen / rocBLAS) on the small-K 1x1 layers at the pool-batch shape."""
import sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from mulactseg_amd import _lib
lib = _lib.load()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for ci, co, h, w in [(128, 64, 256, 512), (64, 256, 256, 512), (256, 64, 256, 512), (256, 128, 256, 512), (128, 512, 128, 256), (512, 128, 128, 256),
                     (64, 64, 192, 192), (64, 256, 192, 192), (256, 64, 192, 192)]:
    x = torch.randn(4, ci, h, w, device='cuda'); wt = torch.randn(co, ci, 1, 1, device='cuda') * 0.05
    wT = wt.reshape(co, ci).t().contiguous()
    y = torch.empty(4, co, h, w, device='cuda')
    ref = F.conv2d(x, wt)
    rc = lib.mas_conv1x1_fwd(x.data_ptr(), wT.data_ptr(), 4, ci, co, h * w, None, None, None, 0, y.data_ptr(), st)
    torch.cuda.synchronize()
    err = float((y - ref).abs().max()) / float(ref.abs().max())
    a = t(lambda: F.conv2d(x, wt)); b = t(lambda: lib.mas_conv1x1_fwd(x.data_ptr(), wT.data_ptr(), 4, ci, co, h * w, None, None, None, 0, y.data_ptr(), st))
    by = 4 * (ci + co) * h * w * 4
    print("1x1 %4d -> %4d @ %3dx%3d: rc %d relerr %.1e | MIOpen %.3f ms | HIP %.3f ms (%.2f TB/s, %.1f TFLOP/s)" % (ci, co, h, w, rc, err, a, b, by / b / 1e9, 2 * 4 * ci * co * h * w / b / 1e9))
