"""Readers of the reference's on-disk formats, on synthetic files; when the reference tree is mounted (build container
only) additionally on its own init data (2 975 Cityscapes images, 6 092 789 valid regions)."""
import json
import os
import pickle
import tempfile

import numpy as np
import pytest
import torch

from mulactseg_amd.dataloader import formats


def test_region_dict_and_datalist_roundtrip():
    tmp = tempfile.mkdtemp()
    rd = {"spx/a.pkl": [6, [2, 4]], "spx/b.pkl": [4, []]}
    with open(os.path.join(tmp, "train.dict"), "w") as f:
        json.dump(rd, f)
    with open(os.path.join(tmp, "train.txt"), "w") as f:
        f.write("img/a.png\tgtFine_dominant/a.png\tspx/a.pkl\nimg/b.png\tgtFine_dominant/b.png\tspx/b.pkl\n")
    ids = formats.load_region_dict(os.path.join(tmp, "train.dict"))
    assert ids == {"spx/a.pkl": [0, 1, 3, 5], "spx/b.pkl": [0, 1, 2, 3]}
    im_idx, suppix = formats.read_datalist(os.path.join(tmp, "train.txt"), "/data", os.path.join(tmp, "train.dict"))
    assert im_idx[0] == ["/data/img/a.png", "/data/gtFine_dominant_ignore/a.png", "/data/spx/a.pkl"]
    assert suppix["/data/spx/b.pkl"] == [0, 1, 2, 3]
    assert formats.id_to_index(os.path.join(tmp, "train.txt")) == {"a": 0, "b": 1}
    explicit = {"spx/a.pkl": [0, 3], "spx/b.pkl": [1, 2]}
    with open(os.path.join(tmp, "e.dict"), "w") as f:
        json.dump(explicit, f)
    assert formats.load_region_dict(os.path.join(tmp, "e.dict")) == explicit
    with open(os.path.join(tmp, "spx.pkl"), "wb") as f:
        pickle.dump({'labels': np.arange(6, dtype=np.int32).reshape(2, 3), 'valid_idxes': [0]}, f)
    assert formats.open_spx(os.path.join(tmp, "spx.pkl")).tolist() == [[0, 1, 2], [3, 4, 5]]
    assert formats.multi_hot_paths("/d", "seeds", 2048)[0] == "/d/superpixel_seed/cityscapes/seeds_2048/train/gtFine_multi_tensor/multi_hot_cls.npy"


def test_multi_hot_from_labels_and_selection_mask():
    spx = np.array([[0, 0, 1, 1], [2, 2, 1, 3]])
    tgt = np.array([[5, 5, 7, 255], [255, 255, 7, 1]])
    cls, size = formats.multi_hot_from_labels(tgt, spx, [0, 1, 2], nseg=5, num_classes=19)
    assert cls[0].nonzero()[0].tolist() == [5] and cls[1].nonzero()[0].tolist() == [7, 19] and cls[2].nonzero()[0].tolist() == [19]
    assert cls[3].sum() == 0 and size.tolist() == [2, 3, 2, -1, -1]
    m = formats.selection_mask(torch.from_numpy(np.array([[0, 5, 1], [3, 2, 5]])), [1, 3], nseg=5)
    assert m.tolist() == [[False, False, True], [True, False, False]]
    assert np.array_equal(m.numpy(), np.isin(np.array([[0, 5, 1], [3, 2, 5]]), [1, 3]))


@pytest.mark.skipif(not os.path.isdir("/root/reference/dataloader/init_data/cityscapes"), reason="reference tree not mounted")
def test_reference_init_data_parses():
    base = "/root/reference/dataloader/init_data/cityscapes"
    ids = formats.load_region_dict(os.path.join(base, "train_seed2048.dict"))
    assert len(ids) == 2975 and sum(len(v) for v in ids.values()) == 6092789
    im_idx, suppix = formats.read_datalist(os.path.join(base, "train_seed2048_or.txt"), "/data/Cityscapes", ids, known_ignore=True)
    assert len(im_idx) == 2975 and im_idx == sorted(im_idx)
    assert len(formats.id_to_index(os.path.join(base, "train_seed2048_or.txt"))) == 2975
