"""A Cityscapes-sized synthetic unlabeled pool that lives on the device (no dataset exists on the GPU box).

``SyntheticPool`` is shaped like the reference's pool dataset (``dataloader/region_cityscapes_or_tensor.py``:
``im_idx`` = list of [image, label, superpixel] paths, ``suppix`` = superpixel path -> ids still in the pool,
``pool[i] -> {'images', 'spx'}``) for 2 975 pictures x 2 048 superpixels.  The superpixel maps are resident as uint16
(2 975 x 4 MiB = 12.5 GB of the 288 GB) and are generated ON the device: the jittered grid of ``synth.superpixel_map``
costs 2 s per picture on a host core, 100 minutes for the pool.

The 499 GB of pool logits cannot be resident, and the acquisition scan does not care where its logits come from, so
``LogitSource`` stands in for ``trainer.net``: ``'images'`` of picture ``i`` is the index ``i`` and ``net(indices)``
returns ``[B, C, H, W]`` logits rolled out of ``nbuf`` resident base pictures (picture ``i`` = base ``i % nbuf`` shifted
by ``17 * (i // nbuf)`` columns), so that every picture has its own region scores while the generator costs one copy.
``duplicates`` makes chosen pictures exact copies of others (same logits, same map): all 2 048 scores of such a pair
tie exactly and only the path rank orders them -- the tie rule of ``active_selection/base.py:37`` at pool scale.

Everything is seeded; the host can fetch any picture's logits / map (``LogitSource.host``, ``SyntheticPool.host_map``)
to feed the CPU oracles the same bytes.
"""
import numpy as np
import torch

from . import synth


def device_superpixel_maps(seeds, H, W, S, device, dtype=torch.int16):
    """[len(seeds), H, W] jittered-grid id maps (the construction of ``synth.superpixel_map``: the displacement field
    is a sum of one-dimensional sines, evaluated per axis and broadcast).  uint16 bits in an int16 tensor."""
    gh, gw = synth.grid_shape(S, H, W)
    ch, cw = H / gh, W / gw
    ay, ax = 0.18 * ch, 0.18 * cw
    y = torch.arange(H, dtype=torch.float64, device=device)
    x = torch.arange(W, dtype=torch.float64, device=device)
    out = torch.empty((len(seeds), H, W), dtype=dtype, device=device)
    for k, seed in enumerate(seeds):
        ph = np.random.RandomState(int(seed)).uniform(0, 2 * np.pi, size=4)
        yy = (y + 0.5 * ay * torch.sin(2 * np.pi * y / (1.9 * ch) + ph[1]))[:, None] + (ay * torch.sin(2 * np.pi * x / (2.7 * cw) + ph[0]))[None, :]
        xx = (x + 0.5 * ax * torch.sin(2 * np.pi * x / (2.3 * cw) + ph[3]))[None, :] + (ax * torch.sin(2 * np.pi * y / (3.1 * ch) + ph[2]))[:, None]
        cy = torch.clamp(torch.floor(yy / ch), 0, gh - 1).to(torch.int32)
        cx = torch.clamp(torch.floor(xx / cw), 0, gw - 1).to(torch.int32)
        out[k] = (cy * gw + cx).to(dtype)
    return out


class LogitSource(torch.nn.Module):
    """Stand-in for the segmentation net of an acquisition round: picture indices in, logits out."""

    def __init__(self, C, H, W, device, nbuf=3, seed=1, alias=None, window=None):
        """``window = (batch_size, first_batch)``: zero-copy mode for timing runs -- the k-th call returns the contiguous
        slice ``base[j : j + len(indices)]`` with ``j = (first_batch + k) % (nbuf - batch_size + 1)`` (picture ``i`` =
        base ``(i // batch_size) % (nbuf - batch_size + 1) + i % batch_size``), sequenced on the host: the scan reads
        resident logits, with no generator copy and no device->host read of the indices in between.  Calls must come in
        pool order (``RegionSelector._iterate`` does)."""
        super().__init__()
        self.C, self.H, self.W, self.nbuf = C, H, W, nbuf
        self.window, self._calls = window, 0
        if window is not None and nbuf < window[0]:
            raise ValueError("window mode needs nbuf >= batch_size")
        self.alias = dict(alias or {})
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        base = 0.35 * torch.randn((nbuf, C, H, W), generator=g, device=device, dtype=torch.float32)
        cm = torch.randint(0, C, (nbuf, 1, H // 32 + 1, W // 32 + 1), generator=g, device=device)
        cm = cm.repeat_interleave(32, 2).repeat_interleave(32, 3)[:, :, :H, :W]
        base.scatter_add_(1, cm, torch.full_like(cm, 0.6, dtype=torch.float32))
        self.base = base.contiguous()

    def _one(self, i, out):
        i = self.alias.get(int(i), int(i))
        shift = (17 * (i // self.nbuf)) % self.W
        src = self.base[i % self.nbuf]
        if shift:
            out[..., shift:] = src[..., :self.W - shift]
            out[..., :shift] = src[..., self.W - shift:]
        else:
            out.copy_(src)

    def forward(self, indices):
        if self.window is not None:
            bs, first = self.window
            j = (first + self._calls) % (self.nbuf - bs + 1)
            self._calls += 1
            return self.base[j:j + indices.shape[0]]
        idx = [int(v) for v in indices.reshape(-1).tolist()]
        z = torch.empty((len(idx), self.C, self.H, self.W), dtype=torch.float32, device=self.base.device)
        for k, i in enumerate(idx):
            self._one(i, z[k])
        return z

    def host(self, i):
        """Logits of picture ``i`` as a numpy array (the bytes the device scan sees)."""
        z = torch.empty((self.C, self.H, self.W), dtype=torch.float32, device=self.base.device)
        self._one(i, z)
        return z.cpu().numpy()


class SyntheticPool(torch.utils.data.Dataset):
    device_resident = True
    suppix_ascending = True      # every suppix list is built (and kept) in ascending id order: RegionActiveDataset may hold them as table rows

    def __init__(self, n_img, H, W, S, device, seed=7, duplicates=None, chunk=64, id_dtype=torch.int16, shard=None):
        """``duplicates``: {picture: picture it copies}.  ``shard`` = (lo, hi): only these pictures' maps are materialised
        (one rank's share of the pool, ``engine.ShardPlan``); the bookkeeping lists always cover the whole pool."""
        self.n_img, self.H, self.W, self.S = n_img, H, W, S
        self.alias = dict(duplicates or {})
        self.lo, self.hi = (0, n_img) if shard is None else (int(shard[0]), int(shard[1]))
        self.maps = torch.empty((max(self.hi - self.lo, 0), H, W), dtype=id_dtype, device=device)
        for lo in range(self.lo, self.hi, chunk):
            hi = min(lo + chunk, self.hi)
            self.maps[lo - self.lo:hi - self.lo] = device_superpixel_maps(
                [seed * 100003 + self.alias.get(i, i) for i in range(lo, hi)], H, W, S, device, id_dtype)
        names = ["city_%05d" % i for i in range(n_img)]
        self.im_idx = [["leftImg8bit/%s.png" % n, "gtFine/%s.png" % n, "superpixel/%s.pkl" % n] for n in names]
        self.suppix = {k[2]: list(range(S)) for k in self.im_idx}
        self._row = {k[2]: i for i, k in enumerate(self.im_idx)}
        self.isselected = np.zeros((n_img, S), dtype=np.uint8)
        self._index = torch.arange(n_img, dtype=torch.int64, device=device)

    def __len__(self):
        return len(self.im_idx)

    def initial_valid_table(self):
        """u8 [n_img, S] of the freshly built pool (every id listed): spares RegionActiveDataset the walk over 6 M list entries."""
        return np.ones((self.n_img, self.S), dtype=np.uint8)

    def __getitem__(self, k):
        i = self._row[self.im_idx[k][2]]
        if not self.lo <= i < self.hi:
            raise IndexError("picture %d is outside this rank's shard [%d, %d)" % (i, self.lo, self.hi))
        return {'images': self._index[i], 'spx': self.maps[i - self.lo]}

    def host_map(self, i):
        return self.maps[i - self.lo].cpu().numpy().view(np.uint16).astype(np.int64)


class SyntheticLabels:
    """The label side of ``RegionActiveDataset``: starts empty, carries ``multi_hot_cls`` [n_img, S, C] u8 (the click cost
    under fair counting, ``region_active_dataset.py:58-65``) and ``id_to_index``."""

    def __init__(self, pool, C, seed=11):
        self.im_idx, self.suppix = [], {}
        rs = np.random.RandomState(seed)
        n, S = pool.n_img, pool.S
        k = rs.choice(4, size=(n, S), p=[0.70, 0.22, 0.06, 0.02]) + 1            # number of classes under a region
        first = rs.randint(0, C, size=(n, S))
        mh = np.zeros((n, S, C), dtype=np.uint8)
        for j in range(4):                                                       # k consecutive classes from a random start
            on = (k > j)
            np.put_along_axis(mh, ((first + j) % C)[..., None], on[..., None].astype(np.uint8), axis=2)
        # put_along_axis overwrites: re-assert the first bit (every region has at least one class)
        np.put_along_axis(mh, first[..., None], 1, axis=2)
        self.multi_hot_cls = mh
        self.id_to_index = {key[2].split('/')[-1].split('.')[0]: i for i, key in enumerate(pool.im_idx)}

    def __len__(self):
        return len(self.im_idx)
