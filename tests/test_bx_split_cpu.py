"""The operand arithmetic of the split-bf16 matrix-core kernels (csrc/bx_split.h), restated in numpy (oracle/bx_split.py): the
three-term split is exact, every term is a bf16 value, and the six partial products of order <= 2 miss a product by at most
2^-23 of it, with either sign -- the size of the rounding an f32 multiply-add chain commits per product (2^-24).  (The kernels
themselves are tested on the GPU against float64 convolutions: tests/test_conv_bx_gpu.py.)"""
import numpy as np

from oracle import bx_split


def _samples():
    rs = np.random.RandomState(0)
    x = np.concatenate([
        rs.standard_normal(200000).astype(np.float32),
        (rs.standard_normal(50000) * 1e-20).astype(np.float32),
        (rs.standard_normal(50000) * 1e20).astype(np.float32),
        rs.randint(-(1 << 24), 1 << 24, 50000).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, np.float32(1) + np.float32(2 ** -23), np.float32(2) - np.float32(2 ** -23), 1.0e38, -1.0e38], np.float32),
    ])
    return x


def test_split_is_exact_and_every_term_is_bf16():
    x = _samples()
    h, m, l = bx_split.split3(x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    for t in (h, m, l):
        assert not np.any(t.view(np.uint32) & np.uint32(0xffff))
    # the terms shrink by 2^-8 each (round to nearest: half an ulp of 8 significand bits)
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(m.astype(np.float64)) <= ax * 2.0 ** -8)
    assert np.all(np.abs(l.astype(np.float64)) <= ax * 2.0 ** -16)
    # next to the subnormal range the third term loses bits (it would be subnormal itself): a bound instead of exactness
    tiny = np.array([1.2e-38, -3.3e-37, 7.0e-36], np.float32)
    th, tm, tl = bx_split.split3(tiny)
    assert np.all(np.abs(th.astype(np.float64) + tm.astype(np.float64) + tl.astype(np.float64) - tiny.astype(np.float64)) < 2.0 ** -133)


def test_six_partial_products_are_within_one_f32_rounding_of_the_product():
    rs = np.random.RandomState(1)
    a = _samples()[:300000]
    b = rs.permutation(_samples())[:300000]
    keep = (np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e37) & (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 1e-30)
    a, b = a[keep], b[keep]
    exact = a.astype(np.float64) * b.astype(np.float64)
    got = bx_split.six_products(a, b)
    rel = np.abs(got - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23 + 2.0 ** -31, rel.max()       # dropped: am*bl + al*bm + al*bl <= 2 (2^-8 2^-16) + 2^-32
    signed = (got - exact) / np.abs(exact)
    assert abs(signed.mean()) < 2.0 ** -30, signed.mean()        # and unbiased (the terms are signed)
    f32 = (a * b).astype(np.float64)                             # one f32 rounding of the same product: up to 2^-24
    assert np.abs(f32 - exact).max() / np.abs(exact).max() <= 2.0 ** -24
    # on small integers everything is exact
    ai = rs.randint(-200, 201, 10000).astype(np.float32)
    bi = rs.randint(-200, 201, 10000).astype(np.float32)
    assert np.array_equal(bx_split.six_products(ai, bi), ai.astype(np.float64) * bi.astype(np.float64))


def test_weight_image_layout():
    """pack_image places w[m, c, tap] where the kernel's A fragment expects it: spot checks of the 1x1 and 3x3 images in both roles."""
    rs = np.random.RandomState(2)
    w1 = rs.standard_normal((130, 40, 1, 1)).astype(np.float32)          # Cout 130 -> 64-row tiles (3), Cin 40 -> two chunks of 32
    img = bx_split.pack_image(w1, 0).reshape(3, 2, 3, 4, 64, 8)
    h, _, _ = bx_split.split3(w1)
    assert img[1, 0, 0, 2, 5, 3] == bx_split.bf16_bits(h)[64 + 5, 16 + 3, 0, 0]      # tile 1 row 5, chunk 0, k = 8 * 2 + 3
    assert img[2, 1, 0, 0, 1, 7] == bx_split.bf16_bits(h)[128 + 1, 32 + 7, 0, 0]     # chunk 1: channel 39
    assert not img[2, :, :, :, 2:, :].any()                                          # rows 130.. are zero
    assert not img[:, 1, :, 1:, :, :].any()                                          # channels 40.. are zero
    w3 = rs.standard_normal((64, 16, 3, 3)).astype(np.float32)
    f = bx_split.pack_image(w3, 0).reshape(1, 2, 3, 10, 64, 8)
    d = bx_split.pack_image(w3, 1).reshape(1, 8, 3, 10, 64, 8)                       # role 1: rows = Cin 16 (one 64-row tile), K = Cout 64
    h3 = bx_split.bf16_bits(bx_split.split3(w3)[0]).reshape(64, 16, 9)
    assert f[0, 1, 0, 4, 9, 2] == h3[9, 8 + 2, 4]                                    # tap 4, channel 10, output channel 9
    assert not f[:, :, :, 9].any()                                                   # the tenth tap is the zero tap
    assert d[0, 3, 0, 2, 5, 1] == h3[24 + 1, 5, 8 - 2]                               # mirrored tap, swapped channel axes


def test_weight_image_layout_of_the_strided_3x3():
    """Role 2: the 3x3 stride-2 weight as 9 x (Cin / 32) chunks of the 1x1 form, tap-major (csrc/conv_bx.hip: nine shifted 1x1 stride-2
    products): w[m, c, tap] sits in chunk tap * (Cin / 32) + c // 32, k group (c % 32) // 8, element c % 8."""
    rs = np.random.RandomState(5)
    w = rs.standard_normal((200, 64, 3, 3)).astype(np.float32)             # Cout 200 -> four 64-row tiles, Cin 64 -> two chunks per tap
    img = bx_split.pack_image(w, 2).reshape(4, 18, 3, 4, 64, 8)
    h, m, l = (bx_split.bf16_bits(t).reshape(200, 64, 9) for t in bx_split.split3(w))
    assert img[1, 7 * 2 + 1, 0, 2, 5, 3] == h[64 + 5, 32 + 16 + 3, 7]        # tile 1 row 5, tap 7, second channel chunk, k = 8 * 2 + 3
    assert img[0, 0, 1, 0, 0, 0] == m[0, 0, 0] and img[3, 17, 2, 3, 7, 7] == l[192 + 7, 63, 8]
    assert not img[3, :, :, :, 8:, :].any()                                # rows 200.. are zero
    assert bx_split.pack_image(w, 2).size * 2 == 4 * 18 * 3 * 4 * 64 * 16   # = mas_conv_bx_packed_bytes(3, 64, 200, 2)
