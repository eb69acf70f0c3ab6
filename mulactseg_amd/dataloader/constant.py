"""Label tables of the two datasets (the public Cityscapes label definition of cityscapesScripts and the PASCAL VOC colour map) --
what the reference keeps in ``dataloader/constant.py:6-89``: ``id_to_train_id`` (raw ``labelIds`` -> the 19 training classes, 255 =
ignored) and the palettes ``decode_target`` paints with."""
import numpy as np

# raw Cityscapes label id -> training id; every id not listed (void, parking, rail track, guard rail, bridge, tunnel, polegroup,
# caravan, trailer, license plate) is ignored
_TRAIN_ID = {7: 0, 8: 1, 11: 2, 12: 3, 13: 4, 17: 5, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13, 27: 14, 28: 15,
             31: 16, 32: 17, 33: 18}
N_RAW_IDS = 34                       # ids 0..33 (+ the license plate, id -1, which the reference reaches as index -1 of its table)

id_to_train_id = np.full(N_RAW_IDS + 1, 255, dtype=np.int64)
for _raw, _train in _TRAIN_ID.items():
    id_to_train_id[_raw] = _train

# the same table for every u8 value (device lookup): values beyond the definition stay 255 -- the reference's 35-entry table raises
# IndexError there, e.g. on a crop padded with 255 BEFORE encoding (region_cityscapes.py:111 after ext_transforms.py:493)
id_to_train_id_u8 = np.full(256, 255, dtype=np.uint8)
id_to_train_id_u8[:N_RAW_IDS] = id_to_train_id[:N_RAW_IDS]

# colour of training id 0..18, then "undefined" (black) and "unselected" (white)
train_id_to_color = np.array([
    (128, 64, 128), (244, 35, 232), (70, 70, 70), (102, 102, 156), (190, 153, 153), (153, 153, 153), (250, 170, 30), (220, 220, 0),
    (107, 142, 35), (152, 251, 152), (70, 130, 180), (220, 20, 60), (255, 0, 0), (0, 0, 142), (0, 0, 70), (0, 60, 100), (0, 80, 100),
    (0, 0, 230), (119, 11, 32), (0, 0, 0), (255, 255, 255)])


def voc_cmap(n=256):
    """The PASCAL VOC colour map: bit k of the class index goes to bit (7 - k // 3) of channel k % 3."""
    idx = np.arange(n)
    cmap = np.zeros((n, 3), dtype=np.uint8)
    for k in range(24):
        cmap[:, k % 3] |= (((idx >> k) & 1) << (7 - k // 3)).astype(np.uint8)
    return cmap


voc_id_to_color_map = np.concatenate([voc_cmap(21), [[255, 255, 255]]]).astype(np.uint8)
