"""Active-set bookkeeping with the semantics of the reference's
``dataloader/region_active_dataset.py:8-105`` (same public methods and pickle layouts).

``expand_training_set`` moves regions from the pool to the labelled set in the order given and stops
after the region that makes the click cost exceed the budget; the consumed prefix is pickled as
``<method>_selection_RR.pkl`` and the lists as ``datalist_RR.pkl``.
"""
import os
import pickle

import numpy as np


class RegionActiveDataset:
    def __init__(self, args, trg_pool_dataset, trg_label_dataset):
        self.args = args
        self.selection_iter = 0
        self.trg_pool_dataset = trg_pool_dataset
        self.trg_label_dataset = trg_label_dataset
        self._valid = None          # u8 [n_img_total, S] mirror of pool.suppix (see pool_valid_mask)
        self._click_cost = None     # (multi_hot_cls it was computed from, u8 [n_img_total, S])
        self._initial_ok = True     # pool.initial_valid_table() describes the lists only until they change behind the table's back

    # -- cost of one region ---------------------------------------------------------------------
    def _fair(self):
        return bool(getattr(self.args, 'fair_counting', False) and getattr(self.args, 'or_labeling', False))

    def _image_index(self, spx_file_path):
        stem = spx_file_path.split('/')[-1].split('.')[0]
        return self.trg_label_dataset.id_to_index[stem]

    def region_cost(self, spx_file_path, suppix_id):
        """Clicks one region costs: number of classes present under fair counting + or-labeling
        (``region_active_dataset.py:58-65``), else 1."""
        if self._fair():
            return int(self.trg_label_dataset.multi_hot_cls[self._image_index(spx_file_path), suppix_id].sum())
        return 1

    def click_cost_table(self):
        """u8 [n_img_total, S]: ``multi_hot_cls[i, s].sum()`` for every region (static over the rounds, computed once),
        or None for unit cost."""
        if not self._fair():
            return None
        mh = self.trg_label_dataset.multi_hot_cls
        if self._click_cost is None or self._click_cost[0] is not mh:
            if hasattr(mh, 'is_cuda'):          # a (device-resident) torch tensor
                tab = mh.sum(dim=2).to(dtype=mh.dtype).cpu().numpy().astype(np.uint8)
            else:
                a = np.asarray(mh)
                if a.dtype == np.uint8:
                    tab = np.einsum('isc->is', a, dtype=np.uint8)       # (2x faster than .sum(axis=2) over the short class axis)
                else:                           # (einsum's 'safe' casting refuses int64 / float label arrays)
                    tab = a.sum(axis=2, dtype=np.uint8)
            self._click_cost = (mh, np.ascontiguousarray(tab))
        return self._click_cost[1]

    def pool_valid_mask(self, nseg):
        """u8 [len(pool.im_idx), nseg] in ``pool.im_idx`` order: 1 where the id is still listed in ``pool.suppix``
        (``active_selection/my_bvsb.py:41-46``).  The table is built once from the lists (or taken from the pool's
        ``initial_valid_table()`` when it offers one), then kept in step by ``expand_training_set``;
        ``load_datalist`` drops it."""
        pool = self.trg_pool_dataset
        if self._valid is None or self._valid.shape[1] != nseg:
            init = getattr(pool, 'initial_valid_table', None)
            tab = init() if (init is not None and self._initial_ok) else None
            if tab is None:
                n_total = len(self.trg_label_dataset.id_to_index)
                tab = np.zeros((n_total, nseg), dtype=np.uint8)
                for key in pool.im_idx:
                    tab[self._image_index(key[2]), pool.suppix[key[2]]] = 1
            self._valid = tab
        rows = np.fromiter((self._image_index(key[2]) for key in pool.im_idx), dtype=np.intp, count=len(pool.im_idx))
        return self._valid[rows]

    # -- selection ------------------------------------------------------------------------------
    def expand_training_set(self, sample_region, selection_count, selection_method):
        """``sample_region``: sorted list of (score, "img,lbl,spx", suppix_id).

        Same end state as the reference loop (:31-73) -- order of ``label.im_idx``, order inside every ``suppix`` list,
        ``isselected``, the pickled prefix -- but without its per-region linear scans: ``key not in label.im_idx`` (:38) is
        answered by a set, the click cost by a table computed once, and ``pool.suppix[path].remove(id)`` (:46, O(S) each) is
        deferred: the ids leaving a list are collected and every touched list is rewritten once, order preserved."""
        pool, label = self.trg_pool_dataset, self.trg_label_dataset
        cost = 0
        n_sup = 0
        listed = {tuple(k) for k in label.im_idx}
        cost_tab = self.click_cost_table()
        has_sel = hasattr(pool, 'isselected')
        leaving = {}            # spx path -> (key, image row, set of ids removed from the pool in this call)
        if self._valid is None:
            self._initial_ok = False        # the lists change now without a table to mirror it: rebuild from the lists later
        for idx, (_, joined, suppix_id) in enumerate(sample_region):
            st = leaving.get(joined)
            if st is None:
                key = joined.split(",")
                spx_path = key[2]
                # membership of an id in the pool list: the valid table answers it when it exists, else a set of the list
                st = leaving[joined] = (key, self._image_index(spx_path), set(), None if self._valid is not None else set(pool.suppix[spx_path]))
                if tuple(key) not in listed:
                    listed.add(tuple(key))
                    label.im_idx.append(key)
                    label.suppix[spx_path] = []
            key, row, gone, present = st
            if (self._valid[row, suppix_id] == 0) if present is None else (suppix_id not in present):
                raise ValueError("list.remove(x): x not in list")            # what pool.suppix[path].remove(id) raises (:46)
            if present is not None:
                present.discard(suppix_id)
            gone.add(suppix_id)
            label.suppix[key[2]].append(suppix_id)
            if has_sel:
                pool.isselected[row, suppix_id] = 1
            if self._valid is not None:
                self._valid[row, suppix_id] = 0
            cost += int(cost_tab[row, suppix_id]) if cost_tab is not None else 1
            n_sup += 1
            if cost > selection_count:
                fname = '%s_selection_%02d.pkl' % (selection_method, self.selection_iter)
                with open(os.path.join(self.args.model_save_dir, fname), "wb") as f:
                    pickle.dump(sample_region[:idx + 1], f)
                break
        emptied = set()
        for key, row, gone, present in leaving.values():
            spx_path = key[2]
            lst = pool.suppix[spx_path]
            if len(gone) < len(lst):
                if len(gone) <= 4:                      # a handful: list.remove keeps the order and runs at C speed
                    for i in gone:
                        lst.remove(i)
                elif not self._delete_by_position(lst, row, gone):
                    pool.suppix[spx_path] = [i for i in lst if i not in gone]
            else:
                pool.suppix.pop(spx_path)
                emptied.add(tuple(key))
        if emptied:
            pool.im_idx[:] = [k for k in pool.im_idx if tuple(k) not in emptied]
        log = getattr(getattr(self.args, 'wandb', None), 'log', None)
        if log is not None and n_sup:
            step = int(getattr(self.args, 'finetune_itrs', 0)) * (self.selection_iter - 1)
            log({"num_selected_spx": n_sup, "num_cls_spx": selection_count / n_sup,
                 "sampling_iter": self.selection_iter}, step=step)
        return n_sup

    def _delete_by_position(self, lst, row, gone):
        """Remove the ids ``gone`` from the list ``lst`` of image ``row`` in place, order preserved, without walking the list in
        Python: in a list that holds its ids in ascending order (how the reference builds them, ``np.unique``; removals keep it)
        an id sits at position (number of listed ids below it), read off the valid table.  The positions are VERIFIED
        (``lst[p] == id`` for every id) before anything is deleted -- deleting verified positions is right whatever the order of the
        rest -- else False: the caller rewrites the list.  (The rewrite, ``[i for i in lst if i not in gone]`` over 2 975 x 2 048
        entries, was 0.15 s of a 0.40 s pool round.)"""
        if self._valid is None:
            return False
        ids = np.fromiter(gone, dtype=np.intp, count=len(gone))
        ids.sort()
        before = self._valid[row].copy()            # (the table already has this call's removals)
        before[ids] = 1
        pos = (np.cumsum(before, dtype=np.intp)[ids] - 1).tolist()
        n = len(lst)
        for p, i in zip(pos, ids.tolist()):
            if p >= n or lst[p] != i:
                return False
        for p in reversed(pos):
            del lst[p]
        return True

    # -- persistence ----------------------------------------------------------------------------
    def dump_datalist(self):
        path = os.path.join(self.args.model_save_dir, 'datalist_%02d.pkl' % self.selection_iter)
        with open(path, "wb") as f:
            pickle.dump({'trg_label_im_idx': self.trg_label_dataset.im_idx,
                         'trg_pool_im_idx': self.trg_pool_dataset.im_idx,
                         'trg_label_suppix': self.trg_label_dataset.suppix,
                         'trg_pool_suppix': self.trg_pool_dataset.suppix}, f)

    def load_datalist(self, datalist_path=None):
        if datalist_path is None:
            datalist_path = os.path.join(self.args.model_save_dir, 'datalist_%02d.pkl' % self.selection_iter)
        with open(datalist_path, "rb") as f:
            data = pickle.load(f)
        self.trg_label_dataset.im_idx = data['trg_label_im_idx']
        self.trg_pool_dataset.im_idx = data['trg_pool_im_idx']
        self.trg_label_dataset.suppix = data['trg_label_suppix']
        self.trg_pool_dataset.suppix = data['trg_pool_suppix']
        self._valid, self._initial_ok = None, False

    def get_trainset(self):
        return self.trg_label_dataset
