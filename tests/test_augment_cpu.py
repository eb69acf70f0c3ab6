"""Training-time geometry: the numpy oracle (oracle/augment.py) and the product's host tables
(dataloader/device_transforms.py) against G9 = real Pillow calls in the reference's order."""
import os
import random

import numpy as np
import pytest

from mulactseg_amd.dataloader import device_transforms as dt
from oracle import augment

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g9_augment.npz")
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def case_inputs(row):
    seed, H, W, ch, cw, nseg = [int(v) for v in row[:6]]
    rs = np.random.RandomState(900 + seed)
    img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    lbl = rs.randint(0, 19, size=(H, W)).astype(np.uint8)
    spx = rs.randint(0, nseg, size=(H, W)).astype(np.int32)
    return seed, H, W, (ch, cw), nseg, img, lbl, spx


def test_oracle_matches_pillow_goldens_and_draw_order():
    g = np.load(GOLD)
    for k, row in enumerate(g['cases']):
        seed, H, W, crop, nseg, img, lbl, spx = case_inputs(row)
        p = augment.draw_params(random.Random(seed), H, W, crop, scale_range=(1.0, 1.0) if seed == 5 else (0.5, 2.0))
        assert [p['th'], p['tw'], p['gap_y'], p['gap_x'], p['i'], p['j'], int(p['flip'])] == [int(v) for v in row[6:]]
        assert dt.draw_params(random.Random(seed), H, W, crop, scale_range=(1.0, 1.0) if seed == 5 else (0.5, 2.0)) == p
        t, (l2, s2) = augment.train_augment(img, [lbl, spx], [255, nseg], p, crop, MEAN, STD)
        assert np.array_equal(t, g['img_%d' % k])                       # float32 bits
        assert np.array_equal(l2, g['lbl_%d' % k].astype(np.int64)) and np.array_equal(s2, g['spx_%d' % k])


@pytest.mark.parametrize("seed", range(12))
def test_host_tables_equal_the_scalar_restatement(seed):
    rs = np.random.RandomState(seed)
    n_in = int(rs.randint(3, 2100))
    n_out = max(1, int(n_in * rs.uniform(0.5, 2.0)))
    b1, k1 = augment.pil_bilinear_coeffs(n_in, n_out)
    b2, k2 = dt.bilinear_tables(n_in, n_out)
    assert np.array_equal(b1, b2) and np.array_equal(k1, k2)
    assert np.array_equal(augment.pil_nearest_index(n_in, n_out), dt.nearest_table(n_in, n_out))


def test_resize_restatement_against_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(3)
    for _ in range(25):
        H, W = int(rs.randint(5, 60)), int(rs.randint(5, 80))
        s = rs.uniform(0.5, 2.0)
        th, tw = max(1, int(H * s)), max(1, int(W * s))
        img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        lab = rs.randint(0, 3000, size=(H, W)).astype(np.int32)
        assert np.array_equal(np.array(Image.fromarray(img).resize((tw, th), Image.BILINEAR)), augment.pil_resize_bilinear_u8(img, th, tw))
        assert np.array_equal(np.array(Image.fromarray(lab).convert('I').resize((tw, th), Image.NEAREST)), augment.pil_resize_nearest(lab, th, tw))
