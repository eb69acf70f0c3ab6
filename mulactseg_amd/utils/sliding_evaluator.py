"""Sliding-window inference -- reference ``utils/sliding_evaluator.py:8-135`` (class scores) and
``utils/sliding_evaluator_plbl.py:8-157`` (class scores + 256-d point features).

The reference moves every window through numpy / cv2 on the host and runs one forward per window.  Here the image
stays on the device: windows are cut as views of the (padded) tensor, forwarded a few at a time as one batch (BN is
in eval mode, so a window's output does not depend on its batch mates) and accumulated into device tensors in the
reference's window order (row-major grid; the last window of a row / column is clamped to the border, so overlaps
differ -- the per-pixel sums use the same addends in the same order).  As in the reference the window SUM is returned
(no division by the coverage count: ``sliding_evaluator.py:117-121`` computes the count but never uses it), cropped
back by the padding margins; its final ``cv2.resize`` to the original size is then always the identity and is omitted.

Returns torch tensors on the model's device ([C,H,W]; the reference returns float64 numpy arrays)."""
import math

import torch
import torch.nn.functional as F


def window_grid(rows, cols, crop, stride_rate):
    """[(y0, x0)] of the reference's window walk over a ``rows`` x ``cols`` (already padded) image
    (``sliding_evaluator.py:96-112``)."""
    ch, cw = crop
    stride_0 = int(math.ceil(ch * stride_rate))
    stride_1 = int(math.ceil(cw * stride_rate))
    r_grid = int(math.ceil((rows - ch) / stride_0)) + 1
    c_grid = int(math.ceil((cols - cw) / stride_1)) + 1
    out = []
    for gy in range(r_grid):
        for gx in range(c_grid):
            e_x = min(gx * stride_1 + cw, cols)
            e_y = min(gy * stride_0 + ch, rows)
            out.append((e_y - ch, e_x - cw))
    return out


def pad_margins(rows, cols, crop):
    """(top, bottom, left, right) zero padding that centres an image smaller than the crop (``:62-74``)."""
    ph = max(crop[0] - rows, 0)
    pw = max(crop[1] - cols, 0)
    return ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2


class SlidingEval(torch.nn.Module):
    with_features = False

    def __init__(self, model, crop_size, stride_rate, device=None, class_number=19, val_id=1, windows_per_forward=4):
        super().__init__()
        self.crop_size = (int(crop_size), int(crop_size)) if not isinstance(crop_size, (tuple, list)) else tuple(map(int, crop_size))
        self.stride_rate = stride_rate
        self.device = device
        self.class_number = class_number
        self.model = model
        if val_id != 1:
            raise NotImplementedError("val_id=2 averages two heads of a model family that is out of scope")
        self.val_id = val_id
        self.feat_dim = 256
        self.windows_per_forward = max(1, int(windows_per_forward))

    def _forward_windows(self, batch):
        """-> (features or None, scores) at window resolution."""
        if self.with_features:
            return self.model.feat_forward(batch)
        return None, self.model(batch)

    @torch.no_grad()
    def forward(self, img):
        img = img.squeeze()
        if img.dim() == 2:
            img = img[None]
        if img.shape[0] < 3:                                  # grey -> three equal channels (:31-35)
            img = img[:1].expand(3, -1, -1)
        rows, cols = img.shape[1:]
        crop = self.crop_size
        top, bottom, left, right = pad_margins(rows, cols, crop)
        padded = F.pad(img, (left, right, top, bottom)) if (top or bottom or left or right) else img
        prow, pcol = padded.shape[1:]
        if max(rows, cols) <= min(crop):
            grid = [(0, 0)]                                   # one padded window (:84-94)
        else:
            grid = window_grid(prow, pcol, crop, self.stride_rate)
        score = torch.zeros((self.class_number, prow, pcol), dtype=torch.float32, device=padded.device)
        feat = torch.zeros((self.feat_dim, prow, pcol), dtype=torch.float32, device=padded.device) if self.with_features else None
        for k in range(0, len(grid), self.windows_per_forward):
            chunk = grid[k:k + self.windows_per_forward]
            batch = torch.stack([padded[:, y:y + crop[0], x:x + crop[1]] for y, x in chunk]).float()
            f, s = self._forward_windows(batch)
            if s.shape[1] > self.class_number:
                s = s[:, :self.class_number]
            for j, (y, x) in enumerate(chunk):                # the reference's accumulation order
                score[:, y:y + crop[0], x:x + crop[1]] += s[j]
                if feat is not None:
                    feat[:, y:y + crop[0], x:x + crop[1]] += f[j]
        score = score[:, top:prow - bottom, left:pcol - right]
        if feat is None:
            return score
        return feat[:, top:prow - bottom, left:pcol - right], score
