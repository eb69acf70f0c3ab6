"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN section 5): numpy restatement of the operand arithmetic of the split-bf16
matrix-core kernels (mulactseg_amd/csrc/bx_split.h, conv_bx.hip:k_bx_pack) -- imported by tests/ only, never by the product.

The kernels compute the f32 convolutions of models/segmentation/backbone/resnet.py:129-160 / deeplabv3.py:85-137 (reference: plain
nn.Conv2d in f32) from an exact three-term bf16 split of every operand:
    x = h + m + l,  h = bf16(x),  m = bf16(x - h),  l = x - h - m          (bf16(): round to nearest even, v_cvt_pk_bf16_f32)
and the six partial products of order <= 2.  This file states the split, the dropped remainder and the weight image bit for bit;
the products themselves are checked against float64 convolutions (the MFMA's internal summation order is not specified)."""
import numpy as np


def rne16(x):
    """round a float32 to the nearest bf16 (ties to even), returned as float32 -- what v_cvt_pk_bf16_f32 does"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7fff) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xffff0000)
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    """(h, m, l) float32 arrays, each exactly representable in bf16 (low 16 bits zero); h + m + l == x exactly for
    2^-100 < |x| < 2^127 (and 0).  |m| <= 2^-8 |x|, |l| <= 2^-16 |x|."""
    x = np.asarray(x, dtype=np.float32)
    h = rne16(x)
    r = (x - h).astype(np.float32)          # exact: at most 16 significand bits
    m = rne16(r)
    l = rne16((r - m).astype(np.float32))   # exact: at most 8 significand bits are left (the conversion changes nothing)
    return h, m, l


def bf16_bits(t):
    """the bf16 bit pattern (uint16) of a float32 array whose low 16 bits are zero"""
    u = np.asarray(t, dtype=np.float32).view(np.uint32)
    assert not np.any(u & np.uint32(0xffff)), "not a bf16 value"
    return (u >> np.uint32(16)).astype(np.uint16)


def six_products(a, b):
    """sum of the six partial products of order <= 2 of a * b, in float64 (every partial product is exact in f32)"""
    ah, am, al = [t.astype(np.float64) for t in split3(a)]
    bh, bm, bl = [t.astype(np.float64) for t in split3(b)]
    return ah * bh + ah * bm + am * bh + ah * bl + al * bh + am * bm


def bx_bm(ksize, M):
    return 128 if (ksize == 1 and M % 128 == 0) else 64


def pack_image(w, role=0, row_scale=None):
    """uint16 image [M tile][chunk][term][k group][row][8] of mas_conv_bx_pack for weight w [Cout, Cin, k, k] (numpy f32).
    role 0: rows = Cout, K = Cin; role 1 (input gradient): rows = Cin, K = Cout, taps mirrored; role 2: the strided 3x3 forward.  row_scale [Cout] (role 0): every
    weight is multiplied by its output channel's entry (one f32 rounding) before the split."""
    w = np.asarray(w, dtype=np.float32)
    if row_scale is not None:
        w = (w * np.asarray(row_scale, dtype=np.float32).reshape(-1, 1, 1, 1)).astype(np.float32)
    Cout, Cin, ks, _ = w.shape
    taps = ks * ks
    wf = w.reshape(Cout, Cin, taps)
    if role == 2:
        # the forward image of the 3x3 STRIDE-2 convolution (csrc/conv_bx.hip: nine shifted 1x1 stride-2 products): the chunks of the
        # 1x1 form in tap-major order, [M tile][tap * (Cin / 32) + channel chunk][term][k group 0..3][row][8], channel = chunk * 32 + 8 g + j
        assert taps == 9 and Cin % 32 == 0
        BM = bx_bm(1, Cout)
        mtiles, cpt = (Cout + BM - 1) // BM, Cin // 32
        val = np.zeros((mtiles, 9 * cpt, 4, BM, 8), dtype=np.float32)
        for mt in range(mtiles):
            rows = np.arange(BM) + mt * BM
            ok = rows < Cout
            for tap in range(9):
                for cc in range(cpt):
                    for g in range(4):
                        for j in range(8):
                            val[mt, tap * cpt + cc, g, ok, j] = wf[rows[ok], cc * 32 + 8 * g + j, tap]
        img = np.zeros((mtiles, 9 * cpt, 3, 4, BM, 8), dtype=np.float32)
        img[:, :, 0], img[:, :, 1], img[:, :, 2] = split3(val)
        return bf16_bits(img).reshape(-1)
    M, K = (Cin, Cout) if role else (Cout, Cin)
    BM = bx_bm(ks, M)
    GA, CK = (4, 32) if taps == 1 else (10, 8)
    mtiles, nch = (M + BM - 1) // BM, (K + CK - 1) // CK
    img = np.zeros((mtiles, nch, 3, GA, BM, 8), dtype=np.float32)
    val = np.zeros((mtiles, nch, GA, BM, 8), dtype=np.float32)
    for mt in range(mtiles):
        for ch in range(nch):
            for g in range(GA):
                for j in range(8):
                    c = ch * CK + (8 * g + j if taps == 1 else j)
                    if c >= K or (taps == 9 and g >= 9):
                        continue
                    tap = 0 if taps == 1 else (8 - g if role else g)
                    rows = np.arange(BM) + mt * BM
                    ok = rows < M
                    if role:
                        val[mt, ch, g, ok, j] = wf[c, rows[ok], tap]
                    else:
                        val[mt, ch, g, ok, j] = wf[rows[ok], c, tap]
    h, m, l = split3(val)
    img[:, :, 0], img[:, :, 1], img[:, :, 2] = h, m, l
    return bf16_bits(img).reshape(-1)
