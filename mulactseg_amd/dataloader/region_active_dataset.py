"""Active-set bookkeeping with the semantics of the reference's
``dataloader/region_active_dataset.py:8-105`` (same public methods and pickle layouts).

``expand_training_set`` moves regions from the pool to the labelled set in the order given and stops
after the region that makes the click cost exceed the budget; the consumed prefix is pickled as
``<method>_selection_RR.pkl`` and the lists as ``datalist_RR.pkl``.
"""
import collections.abc
import os
import pickle
import threading

import numpy as np


class ConsumedPrefix(collections.abc.Sequence):
    """The consumed prefix of a round as ARRAYS -- what the device selection (K4) hands back -- that still behaves as the reference's
    list of ``(score, "img,lbl,spx", suppix_id)`` tuples (``active_selection/base.py:37``: len, indexing, slicing, iteration build
    the tuples on demand), so any ``expand_training_set`` can take it.  This package's ``RegionActiveDataset`` reads the arrays
    instead: the 71 000 tuples of a 100 000-click Cityscapes round cost 33 ms to build and 70 ms to walk one by one.

    ``scores`` f32 [n], ``img`` int [n] (index into ``keys`` = ``pool.im_idx`` as it was when the round was scored), ``ids`` int [n]."""

    def __init__(self, scores, img, ids, keys):
        self.scores = np.asarray(scores)
        self.img = np.asarray(img).astype(np.intp, copy=False)
        self.ids = np.asarray(ids).astype(np.intp, copy=False)
        self.keys = list(keys)          # (a snapshot: expand_training_set edits pool.im_idx)
        if not (len(self.scores) == len(self.img) == len(self.ids)):
            raise ValueError("scores / img / ids differ in length")

    def __len__(self):
        return len(self.ids)

    def tuples(self, n=None):
        """The first ``n`` (default: all) entries as the reference's list of tuples."""
        n = len(self) if n is None else n
        joined = {}
        out = []
        for s, p, i in zip(self.scores[:n].tolist(), self.img[:n].tolist(), self.ids[:n].tolist()):
            j = joined.get(p)
            if j is None:
                j = joined[p] = ','.join(self.keys[p])
            out.append((s, j, i))
        return out

    def __getitem__(self, k):
        if isinstance(k, slice):
            start, stop, step = k.indices(len(self))
            if step == 1 and start == 0:
                return self.tuples(stop)
            return [self[i] for i in range(start, stop, step)]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        return (float(self.scores[k]), ','.join(self.keys[int(self.img[k])]), int(self.ids[k]))


class RegionActiveDataset:
    def __init__(self, args, trg_pool_dataset, trg_label_dataset):
        self.args = args
        self.selection_iter = 0
        self.trg_pool_dataset = trg_pool_dataset
        self.trg_label_dataset = trg_label_dataset
        self._valid = None          # u8 [n_img_total, S] mirror of pool.suppix (see pool_valid_mask)
        self._click_cost = None     # (multi_hot_cls it was computed from, u8 [n_img_total, S])
        self._initial_ok = True     # pool.initial_valid_table() describes the lists only until they change behind the table's back
        self._writer = None         # background thread that writes the selection pickle (wait_for_writes)

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
        self.wait_for_writes()
        if isinstance(sample_region, ConsumedPrefix):
            if self._valid is not None:
                n = self._expand_prefix(sample_region, selection_count, selection_method)
                if n is not None:
                    return n
            sample_region = sample_region.tuples()          # (no table, or an entry the reference loop would raise on: walk it as it does)
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

    # -- the same, from arrays ------------------------------------------------------------------
    def wait_for_writes(self):
        """Join the background write of the last selection pickle (called before anything that reads or rewrites the files)."""
        w, self._writer = self._writer, None
        if w is not None:
            w.join()

    def _expand_prefix(self, sr, selection_count, selection_method):
        """``expand_training_set`` for a ``ConsumedPrefix`` with the valid table in place: the budget cut is one ``cumsum``, the table /
        ``isselected`` updates two fancy assignments, the list edits run once per touched PICTURE (grouped with a stable sort, in the
        order of first appearance -- the order the reference appends to ``label.im_idx``), and the selection pickle is written by a
        background thread.  None when the prefix holds an entry the reference loop would raise on (the caller then walks it tuple by
        tuple, raising where the reference does)."""
        pool, label = self.trg_pool_dataset, self.trg_label_dataset
        n_all = len(sr)
        if n_all == 0:
            return 0
        S = self._valid.shape[1]
        img, ids = sr.img, sr.ids
        if ids.min() < 0 or ids.max() >= S or img.min() < 0 or img.max() >= len(sr.keys):
            return None
        row_of = np.full(len(sr.keys), -1, dtype=np.intp)
        try:
            for p in np.unique(img).tolist():
                row_of[p] = self._image_index(sr.keys[p][2])
        except KeyError:
            return None
        rows = row_of[img]
        cost_tab = self.click_cost_table()
        cost = cost_tab[rows, ids].astype(np.int64) if cost_tab is not None else np.ones(n_all, dtype=np.int64)
        over = np.flatnonzero(np.cumsum(cost) > selection_count)
        n = int(over[0]) + 1 if over.size else n_all
        rows, ids, img = rows[:n], ids[:n], img[:n]
        flat = rows * S + ids
        if not (self._valid[rows, ids] == 1).all() or np.unique(flat).size != n:
            return None                                     # an id that is not in the pool (or twice in the prefix): list.remove would raise
        if over.size:
            fname = os.path.join(self.args.model_save_dir, '%s_selection_%02d.pkl' % (selection_method, self.selection_iter))

            def write(sr=sr, n=n, fname=fname):
                # (every rank of a data-parallel run holds the same prefix and usually the same directory: each writes its own
                #  temporary file and renames it into place -- atomic, so readers never see a torn pickle whoever wins)
                tmp = "%s.tmp.%d" % (fname, os.getpid())
                with open(tmp, "wb") as f:
                    pickle.dump(sr.tuples(n), f)
                os.replace(tmp, fname)
            self._writer = threading.Thread(target=write)
            self._writer.start()
        # where every leaving id sits in its pool list, for all entries at once: in a list that holds its ids in ascending order (how
        # the reference builds them, np.unique; removals keep it) that is the number of listed ids below it -- a row-wise cumsum of
        # the touched rows of the table as it was BEFORE this call.  The positions are verified per picture before anything is deleted.
        urows, ridx = np.unique(rows, return_inverse=True)
        before = self._valid[urows].astype(np.int32)        # (this call's removals are still in: the table is updated below)
        pos = np.cumsum(before, axis=1, dtype=np.int32)[ridx, ids] - 1
        self._valid[rows, ids] = 0
        if hasattr(pool, 'isselected'):
            pool.isselected[rows, ids] = 1
        order = np.argsort(img, kind='stable')              # entries of one picture together, in walk order
        simg, sids, spos = img[order], ids[order], pos[order]
        starts = np.concatenate(([0], np.flatnonzero(np.diff(simg)) + 1))
        ends = np.concatenate((starts[1:], [n]))
        first_seen = order[starts]                          # (stable sort: the first entry of a group is its first appearance)
        listed = {tuple(k) for k in label.im_idx}
        emptied = set()
        for j in np.argsort(first_seen).tolist():
            a, b = int(starts[j]), int(ends[j])
            key = sr.keys[int(simg[a])]
            spx_path = key[2]
            sel = sids[a:b].tolist()
            if tuple(key) not in listed:
                listed.add(tuple(key))
                label.im_idx.append(key)
                label.suppix[spx_path] = []
            label.suppix[spx_path].extend(sel)
            lst = pool.suppix[spx_path]
            if b - a < len(lst):
                ps = spos[a:b].tolist()
                m = len(lst)
                if all(q < m and lst[q] == i for q, i in zip(ps, sel)):     # verified positions: deleting them is right whatever the rest
                    for q in sorted(ps, reverse=True):
                        del lst[q]
                else:                                                        # a list in another order: rewrite it, order preserved
                    gone = set(sel)
                    pool.suppix[spx_path] = [i for i in lst if i not in gone]
            else:
                pool.suppix.pop(spx_path)
                emptied.add(tuple(key))
        if emptied:
            pool.im_idx[:] = [k for k in pool.im_idx if tuple(k) not in emptied]
        log = getattr(getattr(self.args, 'wandb', None), 'log', None)
        if log is not None and n:
            step = int(getattr(self.args, 'finetune_itrs', 0)) * (self.selection_iter - 1)
            log({"num_selected_spx": n, "num_cls_spx": selection_count / n, "sampling_iter": self.selection_iter}, step=step)
        return n

    def _delete_by_position(self, lst, row, gone):
        """Remove the ids ``gone`` from the list ``lst`` of image ``row`` in place, order preserved, without walking the list in
        Python: in a list that holds its ids in ascending order (how the reference builds them, ``np.unique``; removals keep it)
        an id sits at position (number of listed ids below it), read off the valid table.  The positions are VERIFIED
        (``lst[p] == id`` for every id) before anything is deleted -- deleting verified positions is right whatever the order of the
        rest -- else False: the caller rewrites the list.  (The rewrite, ``[i for i in lst if i not in gone]`` over 2 975 x 2 048
        entries, was 0.15 s of a 0.40 s pool round.)"""
        if self._valid is None:
            return False
        ids = np.sort(gone) if isinstance(gone, np.ndarray) else np.sort(np.fromiter(gone, dtype=np.intp, count=len(gone)))
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
        self.wait_for_writes()
        path = os.path.join(self.args.model_save_dir, 'datalist_%02d.pkl' % self.selection_iter)
        with open(path, "wb") as f:
            pickle.dump({'trg_label_im_idx': self.trg_label_dataset.im_idx,
                         'trg_pool_im_idx': self.trg_pool_dataset.im_idx,
                         'trg_label_suppix': self.trg_label_dataset.suppix,
                         'trg_pool_suppix': self.trg_pool_dataset.suppix}, f)

    def load_datalist(self, datalist_path=None):
        self.wait_for_writes()
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
