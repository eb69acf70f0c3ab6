"""mas_train_augment (csrc/augment.hip) == the numpy oracle == Pillow goldens, bit for bit (floats included)."""
import random

import numpy as np
import pytest
import torch

from test_augment_cpu import GOLD, MEAN, STD, case_inputs

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mulactseg_amd.dataloader import device_transforms
    return device_transforms


def test_goldens_bit_exact():
    dt = _gpu()
    g = np.load(GOLD)
    for k, row in enumerate(g['cases']):
        seed, H, W, crop, nseg, img, lbl, spx = case_inputs(row)
        aug = dt.DeviceTrainAugment(size=crop, scale_range=(1.0, 1.0) if seed == 5 else (0.5, 2.0), pad_values=[255, nseg],
                                    rng=random.Random(seed))
        t, (l2, s2) = aug(torch.from_numpy(img).cuda(), [torch.from_numpy(lbl).cuda(), torch.from_numpy(spx).cuda()])
        assert np.array_equal(t.cpu().numpy(), g['img_%d' % k])
        assert l2.dtype == torch.uint8 and np.array_equal(l2.cpu().numpy(), g['lbl_%d' % k])
        assert s2.dtype == torch.int64 and np.array_equal(s2.cpu().numpy(), g['spx_%d' % k])


@pytest.mark.parametrize("seed,H,W,crop", [(11, 256, 512, (192, 192)), (12, 256, 512, (192, 192)), (13, 130, 75, (96, 128)),
                                           (14, 1024, 2048, (768, 768))])
def test_matches_oracle_on_larger_shapes(seed, H, W, crop):
    dt = _gpu()
    from oracle import augment
    rs = np.random.RandomState(seed)
    img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    spx = rs.randint(0, 2048, size=(H, W)).astype(np.int64)
    lbl = rs.randint(0, 20, size=(H, W)).astype(np.uint8)
    p = dt.draw_params(random.Random(seed), H, W, crop)
    aug = dt.DeviceTrainAugment(size=crop, pad_values=[255, 2048], keep_u8=False)
    t, (l2, s2) = aug(torch.from_numpy(img).cuda(), [torch.from_numpy(lbl).cuda(), torch.from_numpy(spx).cuda()], params=p)
    te, (le, se) = augment.train_augment(img, [lbl, spx], [255, 2048], p, crop, MEAN, STD)
    assert np.array_equal(t.cpu().numpy(), te)
    assert np.array_equal(l2.cpu().numpy(), le) and np.array_equal(s2.cpu().numpy(), se)


def test_rejects_host_tensors():
    dt = _gpu()
    with pytest.raises(ValueError):
        dt.DeviceTrainAugment()(torch.zeros((8, 8, 3), dtype=torch.uint8), [])


def test_resident_region_dataset_samples():
    """dataloader/resident.py: sample dictionary of the reference's region dataset from resident tensors; the mask is
    np.isin(spx, selected ids) and the pad id is never selected; pool items are the normalised full picture."""
    _gpu()
    import types
    from mulactseg_amd.dataloader.resident import ResidentRegionDataset
    rs = np.random.RandomState(5)
    nseg, n = 40, 3
    pics = [torch.from_numpy(rs.randint(0, 256, size=(96, 160, 3)).astype(np.uint8)).cuda() for _ in range(n)]
    spxs = [torch.from_numpy(rs.randint(0, nseg, size=(96, 160)).astype(np.int16)).cuda() for _ in range(n)]
    mh = torch.from_numpy(rs.randint(0, 2, size=(n, nseg, 20)).astype(np.uint8)).cuda()
    names = [("img%d.png" % i, "lbl%d.png" % i, "spx%d.pkl" % i) for i in range(n)]
    args = types.SimpleNamespace(nseg=nseg, ignore_idx=255)
    region = {"spx0.pkl": [1, 5, 7], "spx2.pkl": [0, 39]}
    ds = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-label', region_dict=region, rng=random.Random(3))
    ds.transform.size = (64, 64)
    assert len(ds) == 2 and tuple(ds.im_idx[1]) == names[2]
    for idx in range(2):
        s = ds[idx]
        assert tuple(s['images'].shape) == (3, 64, 64) and s['images'].dtype == torch.float32
        assert s['spx'].dtype == torch.int64 and tuple(s['spmask'].shape) == (64, 64)
        sel = region[s['fnames'][2]]
        assert np.array_equal(s['spmask'].cpu().numpy(), np.isin(s['spx'].cpu().numpy(), sel))
        assert not bool(s['spmask'][s['spx'] == nseg].any())
        assert torch.equal(s['labels'], mh[ds.names[s['fnames'][2]]])
    pool = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-ulabel')
    item = pool[1]
    ref = pics[1].cpu().numpy().transpose(2, 0, 1).astype(np.float32) / np.float32(255)       # true division, as to_tensor on the CPU
    ref = (ref - np.asarray(MEAN, np.float32)[:, None, None]) / np.asarray(STD, np.float32)[:, None, None]
    assert np.array_equal(item['images'].cpu().numpy(), ref) and torch.equal(item['spx'], spxs[1].long())
    assert sorted(pool.suppix["spx1.pkl"]) == list(range(nseg))


def test_resident_provider_batches_forever():
    _gpu()
    import types
    from mulactseg_amd.dataloader import ResidentProvider
    from mulactseg_amd.dataloader.resident import ResidentRegionDataset
    rs = np.random.RandomState(9)
    nseg, n = 20, 5
    pics = [torch.from_numpy(rs.randint(0, 256, size=(80, 96, 3)).astype(np.uint8)).cuda() for _ in range(n)]
    spxs = [torch.from_numpy(rs.randint(0, nseg, size=(80, 96)).astype(np.int32)).cuda() for _ in range(n)]
    mh = torch.from_numpy(rs.randint(0, 2, size=(n, nseg, 20)).astype(np.uint8)).cuda()
    names = [("i%d" % i, "l%d" % i, "s%d" % i) for i in range(n)]
    ds = ResidentRegionDataset(types.SimpleNamespace(nseg=nseg, ignore_idx=255), pics, spxs, mh, names, split='active-label',
                               region_dict={"s%d" % i: [i, i + 1] for i in range(n)}, rng=random.Random(1))
    ds.transform.size = (48, 48)
    prov = ResidentProvider(ds, batch_size=2, drop_last=True, shuffle=True, rng=random.Random(2), prefetch=False)
    assert len(prov) == 2
    seen, batches = [], []
    for _ in range(5):
        b = next(prov)
        assert tuple(b['images'].shape) == (2, 3, 48, 48) and b['images'].is_cuda and b['spmask'].dtype == torch.bool
        assert tuple(b['labels'].shape) == (2, nseg, 20) and len(b['fnames']) == 2
        seen += [f[2] for f in b['fnames']]
        batches.append(b)
    assert prov.epoch == 2 and prov.iteration == 5 and len(set(seen)) == n
    # prefetching on a side stream yields exactly the same batches (same order of the random draws)
    ds.transform.rng = random.Random(1)
    pre = ResidentProvider(ds, batch_size=2, drop_last=True, shuffle=True, rng=random.Random(2), prefetch=True)
    for b in batches:
        c = next(pre)
        assert c['fnames'] == b['fnames'] and torch.equal(c['images'], b['images']) and torch.equal(c['spx'], b['spx'])
        assert torch.equal(c['spmask'], b['spmask'])
    assert pre.epoch == 2 and pre.iteration == 5
