"""The two packed-f32 operand-select forms the library writes by hand (common.h: mas_pk_mul_lo / mas_pk_mul_hi -- the half-select on SRC0) are
right beside another kernel's MFMA waves, where the src1 form is not (NOTEBOOK.md section 16.7, tools/pk_opsel_probe.py).  The probe kernels
live in the test-support library (tests/libmulactseg_test.so)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib_and_neighbours():
    from mulactseg_amd import _lib, ops
    from helpers import _test_lib
    lib = _test_lib()
    lib.mas_test_pk_opsel.restype = ctypes.c_int
    lib.mas_test_pk_opsel.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.mas_test_unit_busy.restype = ctypes.c_int
    lib.mas_test_unit_busy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    conv = torch.nn.Conv2d(512, 512, 3, padding=2, dilation=2, bias=False).cuda()
    x = torch.randn((4, 512, 32, 64), device='cuda')

    def conv_bx(st):
        with torch.no_grad():
            for _ in range(30):
                ops.conv_bx(conv, x)

    def mfma_pair(st):
        _lib.check(lib.mas_test_unit_busy(11, 1024, 20000, None, st.cuda_stream), "mas_test_unit_busy")
    return _lib, lib, {"k_conv_bx": conv_bx, "two back-to-back MFMAs": mfma_pair}


def _wrong(_lib, lib, mode, neighbour, reps=3):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = torch.zeros(64, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    for _ in range(reps):
        with torch.cuda.stream(sb):
            neighbour(sb)
        with torch.cuda.stream(sa):
            _lib.check(lib.mas_test_pk_opsel(mode, 2048, 2000, bad.data_ptr(), sa.cuda_stream), "mas_test_pk_opsel")
        with torch.cuda.stream(sb):
            neighbour(sb)
        torch.cuda.synchronize()
    return bad.cpu().view(4, 16).sum(dim=1).tolist()


@pytest.mark.parametrize("mode,form", [(2, "v_pk_mul_f32 d, a, b op_sel_hi:[0,1]  (mas_pk_mul_lo)"), (4, "v_pk_mul_f32 d, a, b op_sel:[1,0]  (mas_pk_mul_hi)"),
                                       (3, "v_pk_mul_f32 d, a, b"), (0, "v_pk_mul_f32 d, a, b op_sel_hi:[1,0]")])
def test_the_operand_select_forms_the_library_uses_are_right_beside_mfma_neighbours(mode, form):
    _lib, lib, neighbours = _lib_and_neighbours()
    for name, nb in neighbours.items():
        nb(torch.cuda.current_stream())
        torch.cuda.synchronize()
        assert _wrong(_lib, lib, mode, nb) == [0, 0, 0, 0], (form, name)


def test_report_the_src1_form_beside_mfma_neighbours():
    """Not an assertion about the hardware: prints what this box does with the form the library avoids (wrong results by lane quarter)."""
    _lib, lib, neighbours = _lib_and_neighbours()
    for name, nb in neighbours.items():
        q = _wrong(_lib, lib, 1, nb)
        print("v_pk_mul_f32 d, a, b op_sel:[0,1] beside %s: wrong results by lane quarter %s" % (name, q))
        assert len(q) == 4              # (informational: the library does not contain the form, whatever a box does with it)
