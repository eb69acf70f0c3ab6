"""Developer probe: MIOpen's time for the 1x1 convolutions of the backbone at the pool-batch shape [4,3,1024,2048] against
the time their compulsory HBM traffic would take at 5 TB/s."""
import time, torch
import torch.nn.functional as F
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
shapes = [  # (Cin, Cout, H, W, count per forward)
    (128, 64, 256, 512, 1), (64, 256, 256, 512, 3), (256, 64, 256, 512, 2), (256, 128, 256, 512, 1), (128, 512, 128, 256, 4),
    (512, 128, 128, 256, 3), (512, 256, 128, 256, 1), (256, 1024, 64, 128, 6), (1024, 256, 64, 128, 5), (1024, 512, 64, 128, 1),
    (512, 2048, 64, 128, 3), (2048, 512, 64, 128, 2), (2048, 256, 64, 128, 5)]
tot = ideal = 0.0
with torch.no_grad():
    for ci, co, h, w, cnt in shapes:
        x = torch.randn(4, ci, h, w, device='cuda'); wt = torch.randn(co, ci, 1, 1, device='cuda') * 0.05
        ms = t(lambda: F.conv2d(x, wt))
        by = 4 * (ci + co) * h * w * 4
        fl = 2 * 4 * ci * co * h * w
        print("1x1 %4d -> %4d @ %3dx%3d  x%d : %.3f ms  (%.1f TFLOP/s, %.2f TB/s; HBM-ideal %.3f ms)" % (ci, co, h, w, cnt, ms, fl / ms / 1e9, by / ms / 1e9, by / 5e9))
        tot += ms * cnt; ideal += by / 5e9 * cnt
print("sum over a forward: %.2f ms, HBM-ideal %.2f ms" % (tot, ideal))
