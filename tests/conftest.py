import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- MAS_TEST_NEIGHBOUR=1: every GPU test runs while a second stream keeps this package's MFMA convolution kernel on the chip ---------
# (round 6, NOTEBOOK.md section 16.7: an instruction form that is right alone was wrong beside another kernel's matrix-core waves.  The
# parity tests compare against the oracle, so running them beside such a neighbour checks every kernel for that kind of sensitivity.
# Off by default: it triples the run time and the timing-based tests are not meant for it.)
class _Neighbour:
    def __init__(self):
        import threading
        import torch
        from mulactseg_amd import ops
        self.torch, self.ops = torch, ops
        self.stream = torch.cuda.Stream()
        with torch.random.fork_rng(devices=[torch.cuda.current_device()]), torch.cuda.stream(self.stream):    # (the tests' random streams stay theirs)
            self.conv = torch.nn.Conv2d(512, 512, 3, padding=2, dilation=2, bias=False).cuda()
            self.x = torch.randn((4, 512, 32, 64), device='cuda')
        self.stream.synchronize()
        self.stop = threading.Event()
        self.launched = 0
        self.thread = threading.Thread(target=self.run, daemon=True)
        self.thread.start()

    def run(self):
        torch = self.torch
        with torch.cuda.stream(self.stream), torch.no_grad():
            while not self.stop.is_set():
                for _ in range(16):
                    self.ops.conv_bx(self.conv, self.x)
                self.launched += 16
                ev = torch.cuda.Event()
                ev.record(self.stream)
                ev.synchronize()                     # at most one group of launches queued: the neighbour ends with the test

    def close(self):
        self.stop.set()
        self.thread.join(timeout=30)
        self.stream.synchronize()


@pytest.fixture(autouse=True)
def _mfma_neighbour(request):
    if os.environ.get("MAS_TEST_NEIGHBOUR") != "1" or request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    nb = _Neighbour()
    try:
        yield
    finally:
        nb.close()
        assert nb.launched > 0
