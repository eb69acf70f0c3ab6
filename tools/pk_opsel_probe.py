#!/usr/bin/env python
"""Which packed-f32 operand-select forms return wrong results in lanes 48-63 while a workgroup of ANOTHER kernel shares the compute unit.

Found in round 6 (NOTEBOOK.md section 16.7): `k_single_pass<..., LOWRES, X4>` miscounted next to `k_conv_bx` on a second stream.  The probe
kernel (mulactseg_amd/csrc/test_support.hip: k_test_pk_opsel -- test infrastructure, tests/libmulactseg_test.so) evaluates ONE instruction
form per launch on pseudo-random operands against scalar instructions and counts the differing results per lane; the neighbours keep one
kind of unit busy (matrix cores / LDS / vector ALUs) or are this package's convolution kernel.

    python tools/pk_opsel_probe.py            # table on stdout and in gpurun_out/pk_opsel_probe.md
    PK_FORMS=1,3 PK_MORE=1 python tools/pk_opsel_probe.py      # only these instruction forms; more single-instruction neighbours"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FORMS = {0: "v_pk_mul_f32 d, a, b op_sel_hi:[1,0]", 1: "v_pk_mul_f32 d, a, b op_sel:[0,1]", 2: "v_pk_mul_f32 d, a, b op_sel_hi:[0,1]",
         3: "v_pk_mul_f32 d, a, b", 4: "v_pk_mul_f32 d, a, b op_sel:[1,0]", 5: "v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]",
         6: "v_pk_add_f32 d, a, b op_sel:[0,1]", 7: "v_pk_fma_f32 d, a, b, c op_sel:[1,0,0]", 8: "v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]",
         9: "v_pk_fma_f32 d, a, b, c op_sel:[0,0,1]", 10: "v_pk_fma_f32 d, a, s[n:n+1], c op_sel:[0,1,0]",
         11: "v_pk_mul_f32 d, s[n:n+1], b op_sel:[1,0]", 12: "v_pk_mov_b32 d, a, b op_sel:[1,0]",
         13: "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]"}

# neighbours from the test-support library: (kind code of mas_test_unit_busy, trips).  Every MFMA neighbour accumulates in AGPRs.
BUSY = {'2 MFMAs back to back per trip': (11, 20000), '2 MFMAs per trip, scalar instructions between': (0, 20000),
        '4 MFMAs back to back behind a ds_read_b128': (6, 8000), 'ds_read_b128': (1, 40000), 'v_pk_fma_f32': (2, 100000)}
MORE = {'v_permlane16_swap': (3, 100000), 'v_add_f32_dpp': (4, 60000), 'v_cvt_pk_bf16_f32': (5, 100000), 'v_permlane32_swap': (7, 100000),
        'global_load_dwordx4': (8, 20000), 'global_store_dword': (9, 20000), 'ds_write_b128': (10, 60000), 's_barrier': (12, 100000)}
REAL = ('k_conv_bx 1x1', 'k_wgrad_bx (1x1)', 'k_wgrad_bx3 (3x3)', 'conv_mfma 3x3', 'rocBLAS mm', 'scan x4')


def main():
    from mulactseg_amd import _lib, ops
    from tests.helpers import _test_lib
    dev = torch.device('cuda:0')
    lib = _test_lib()
    lib.mas_test_pk_opsel.restype = ctypes.c_int
    lib.mas_test_pk_opsel.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.mas_test_unit_busy.restype = ctypes.c_int
    lib.mas_test_unit_busy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    src = torch.randn((1 << 18,), device=dev)
    c3 = torch.nn.Conv2d(512, 512, 3, padding=2, dilation=2, bias=False).to(dev)
    x3 = torch.randn((4, 512, 32, 64), device=dev)
    c1 = torch.nn.Conv2d(1024, 256, 1, bias=False).to(dev)
    x1 = torch.randn((4, 1024, 32, 64), device=dev)
    dy1 = torch.randn((4, 256, 32, 64), device=dev)
    dy3 = torch.randn((4, 512, 32, 64), device=dev)
    big = torch.randn((4096, 4096), device=dev)
    zq = torch.randn((4, 20, 128, 256), device=dev)
    spx = torch.randint(0, 64, (4, 512, 1024), device=dev, dtype=torch.int32)

    def neighbour(kind, st):
        if kind == 'k_conv_bx':
            with torch.no_grad():
                for _ in range(40):
                    ops.conv_bx(c3, x3)
        elif kind == 'k_conv_bx 1x1':
            with torch.no_grad():
                for _ in range(80):
                    ops.conv_bx(c1, x1)
        elif kind == 'k_wgrad_bx (1x1)':
            for _ in range(40):
                ops.conv_wgrad(x1, dy1, 1, 1, 1)
        elif kind == 'k_wgrad_bx3 (3x3)':
            for _ in range(20):
                ops.conv_wgrad(x3, dy3, 3, 1, 2)
        elif kind == 'conv_mfma 3x3':
            with torch.no_grad():
                for _ in range(40):
                    ops.conv_mfma(c3, x3)
        elif kind == 'rocBLAS mm':
            for _ in range(8):
                torch.mm(big, big)
        elif kind == 'scan x4':
            for _ in range(20):
                ops.single_pass_accum_lowres(zq, (512, 1024), spx, 64, ops.inv_temperature(0.1))
        elif kind in BUSY:
            code, iters = BUSY[kind]
            _lib.check(lib.mas_test_unit_busy(code, 1024, iters, src.data_ptr(), st.cuda_stream), "busy")

    A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    if os.environ.get('PK_MORE'):
        BUSY.update(MORE)
    kinds = ('none', 'k_conv_bx') + REAL + tuple(BUSY)
    only = os.environ.get('PK_FORMS')
    for k in kinds:
        neighbour(k, torch.cuda.current_stream())
    torch.cuda.synchronize()
    reps, blocks, iters = 3, 2048, 2000
    total = reps * blocks * 256 * iters * 2
    rows = []
    for mode in (sorted(FORMS) if not only else [int(v) for v in only.split(',')]):
        row = []
        for kind in kinds:
            bad = torch.zeros(64, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for _ in range(reps):
                with torch.cuda.stream(B):
                    neighbour(kind, B)
                with torch.cuda.stream(A):
                    _lib.check(lib.mas_test_pk_opsel(mode, blocks, iters, bad.data_ptr(), A.cuda_stream), "probe")
                with torch.cuda.stream(B):
                    neighbour(kind, B)
                torch.cuda.synchronize()
            row.append(bad.cpu().view(4, 16).sum(dim=1).tolist())
        rows.append((mode, row))
        print("%-52s %s" % (FORMS[mode], "  ".join("%s: %s" % (k, q) for k, q in zip(kinds, row))), flush=True)
    out = ["# packed-f32 operand select next to another kernel's workgroups on the same CU (MI355X, gfx950)", "",
           "command: `python tools/pk_opsel_probe.py` -- wrong results (of %d per cell) in lanes 48-63 (a list of the four lane quarters where any other lane was wrong: none was); the probe "
           "kernel (tests/libmulactseg_test.so: k_test_pk_opsel, one instruction form against scalar instructions on pseudo-random operands) runs on "
           "one stream, the neighbour on another.  a, b, c, d: VGPR pairs." % total, "",
           "| instruction form | " + " | ".join(kinds) + " |", "|---|" + "---|" * len(kinds)]
    for mode, row in rows:
        out.append("| `%s` | %s |" % (FORMS[mode], " | ".join(str(q[3]) if sum(q[:3]) == 0 else str(q) for q in row)))
    path = os.path.join(ROOT, "gpurun_out", "pk_opsel_probe.md")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
