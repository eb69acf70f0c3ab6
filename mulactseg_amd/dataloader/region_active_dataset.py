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


def _atomic_pickle(fname, obj):
    """Write to a temporary file beside the target and rename it into place: readers never see a torn pickle; a failed write leaves
    no temporary behind."""
    tmp = "%s.tmp.%d" % (fname, os.getpid())
    try:
        with open(tmp, "wb") as f:
            pickle.dump(obj, f)
        os.replace(tmp, fname)
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise


class _Writer:
    """The selection pickle, written by a background thread (it is off the round's critical path: nothing reads it before the next
    ``wait_for_writes``).  ``join`` re-raises what the write raised."""

    def __init__(self, fname, make):
        self.error = None

        def run():
            try:
                _atomic_pickle(fname, make())
            except BaseException as e:       # noqa: B902 -- handed to the caller of join()
                self.error = e
        self.thread = threading.Thread(target=run)
        self.thread.start()

    def join(self):
        self.thread.join()
        if self.error is not None:
            raise self.error


class _FromTable:
    """A pool list that nobody has looked at since it was last known to be in ascending order: it IS the set bits of its row of
    the valid table (``RegionActiveDataset._valid``, edited in place by every round), so it is built when it is asked for."""
    __slots__ = ("table", "row")

    def __init__(self, table, row):
        self.table, self.row = table, row

    def build(self):
        return np.flatnonzero(self.table[self.row]).tolist()


class _Appended:
    """A label list = what it was (a real list, or nothing) + the id runs the rounds since then selected, in selection order."""
    __slots__ = ("base", "runs")

    def __init__(self, base=None):
        self.base, self.runs = base, []

    def build(self):
        out = self.base if self.base is not None else []
        for r in self.runs:
            out.extend(r.tolist())
        return out


class LazySuppix(dict):
    """``{superpixel path: list of ids}`` -- the reference's ``suppix`` dictionaries (``dataloader/region_active_dataset.py:16-80``
    edits them entry by entry) -- whose lists are BUILT ON FIRST ACCESS from the arrays ``RegionActiveDataset`` keeps.  A 100 000-click
    Cityscapes round touches all 2 975 lists; the training loader of the next round reads the lists of the pictures it samples, the
    next acquisition round reads none (it uses the valid table), ``dump_datalist`` reads all of them once.  Everything a ``dict``
    offers works and returns real lists (``[]``, ``get``, ``pop``, ``items``, ``values``, ``==``, ``copy``, pickling -- as a plain
    ``dict`` of lists, the reference's ``datalist_RR.pkl`` layout); a list that has been handed out is a real list from then on, and
    later rounds edit it in place as the reference does."""

    def _raw(self, key):
        return dict.__getitem__(self, key)

    def __iter__(self):
        # (defined so that dict(lazy), {**lazy} and dict.update(lazy) do not take CPython's raw-slot fast path for exact-dict
        #  iteration: with an overridden __iter__ they go through keys() and __getitem__, i.e. they see real lists)
        return dict.__iter__(self)

    def keys(self):
        return dict.keys(self)

    def _real(self, key, v):
        if isinstance(v, (_FromTable, _Appended)):
            v = v.build()
            dict.__setitem__(self, key, v)
        return v

    def __getitem__(self, key):
        return self._real(key, dict.__getitem__(self, key))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        if key in self:
            v = self[key]
            dict.__delitem__(self, key)
            return v
        if default:
            return default[0]
        raise KeyError(key)

    def popitem(self):
        k = next(reversed(self))
        return k, self.pop(k)

    def setdefault(self, key, default=None):
        if key not in self:
            dict.__setitem__(self, key, default)
        return self[key]

    def materialize(self):
        for k in dict.keys(self):
            self[k]
        return self

    def values(self):
        return dict.values(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def copy(self):
        return dict(self.items())

    def __eq__(self, other):
        return dict.__eq__(self.materialize(), other.materialize() if isinstance(other, LazySuppix) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __reduce_ex__(self, protocol):
        return (dict, (), None, None, iter(self.items()))       # pickles (and deep-copies) as a plain dict of lists

    def __repr__(self):
        return dict.__repr__(self.materialize())

    def pending(self):
        """How many lists have not been built (tests, the bench's breakdown)."""
        return sum(1 for v in dict.values(self) if isinstance(v, (_FromTable, _Appended)))


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
                    # (a [rows, classes] reduction over the short, contiguous class axis: torch's CPU kernel -- vectorised, several
                    #  threads -- takes 4-12 ms for a Cityscapes-sized table, 116 MB of labels; numpy's einsum 17-55 ms, .sum(axis=2) twice that)
                    if a.flags.c_contiguous and a.flags.writeable and a.ndim == 3:
                        import torch
                        tab = torch.from_numpy(a).view(-1, a.shape[2]).sum(dim=1, dtype=torch.uint8).view(a.shape[0], a.shape[1]).numpy()
                    else:
                        tab = np.einsum('isc->is', a, dtype=np.uint8)
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
            canonical = tab is not None and bool(getattr(pool, 'suppix_ascending', False))
            if tab is None:
                n_total = len(self.trg_label_dataset.id_to_index)
                tab = np.zeros((n_total, nseg), dtype=np.uint8)
                canonical = True
                for key in pool.im_idx:
                    ids = np.asarray(pool.suppix[key[2]], dtype=np.intp)
                    tab[self._image_index(key[2]), ids] = 1
                    # (ascending and duplicate-free -- how the reference builds its lists, np.unique, and removals keep it -- : the list
                    #  is then exactly the set bits of its table row)
                    canonical = canonical and (ids.size < 2 or bool((ids[1:] > ids[:-1]).all()))
            self._valid = tab
            rows = np.fromiter((self._image_index(key[2]) for key in pool.im_idx), dtype=np.intp, count=len(pool.im_idx))
            if canonical and os.environ.get("MAS_LAZY_LISTS", "on") != "off":
                self._install_lazy_lists(rows)
        else:
            rows = np.fromiter((self._image_index(key[2]) for key in pool.im_idx), dtype=np.intp, count=len(pool.im_idx))
        return self._valid[rows]

    def _install_lazy_lists(self, rows):
        """Swap ``pool.suppix`` / ``label.suppix`` for ``LazySuppix`` mappings: every pool list (just checked to be ascending) becomes
        "the set bits of its table row, built when asked for", label lists keep what they hold and take appended id runs.  The
        array path of ``expand_training_set`` then edits the table and appends arrays; Python lists appear where somebody reads them."""
        pool, label = self.trg_pool_dataset, self.trg_label_dataset
        if not isinstance(pool.suppix, dict) or not isinstance(label.suppix, dict):
            return
        lazy = LazySuppix()
        for key, row in zip(pool.im_idx, rows.tolist()):      # (rows: the table row of every entry of im_idx, already looked up)
            dict.__setitem__(lazy, key[2], _FromTable(self._valid, row))
        if len(lazy) != len(pool.suppix):       # (entries without a picture in im_idx: leave the dictionaries alone)
            return
        old, pool.suppix = pool.suppix, lazy
        if not isinstance(label.suppix, LazySuppix):
            label.suppix = LazySuppix(label.suppix)
        # The lists just replaced are 6 M Python ints for a Cityscapes pool: freeing them takes 50 ms of interpreter time.  A daemon
        # thread drops them entry by entry (a bytecode boundary between any two, so the round's own thread -- mostly waiting for the
        # device at this point -- is never held up for longer than one list).
        if len(old) > 64:
            def reap(d=old):
                while d:
                    d.popitem()
            threading.Thread(target=reap, daemon=True).start()
        del old

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
                if self._writes_files():
                    fname = '%s_selection_%02d.pkl' % (selection_method, self.selection_iter)
                    _atomic_pickle(os.path.join(self.args.model_save_dir, fname), sample_region[:idx + 1])
                break
        emptied = set()
        psup = pool.suppix
        for key, row, gone, present in leaving.values():
            spx_path = key[2]
            if isinstance(psup, LazySuppix) and isinstance(psup._raw(spx_path), _FromTable):
                # a list nobody has built: it IS its row of the valid table, which the loop above has already edited (building it
                # here would give the list AFTER the removals, and `gone` would be compared with what is left)
                if not self._valid[row].any():
                    dict.__delitem__(psup, spx_path)
                    emptied.add(tuple(key))
                continue
            lst = psup[spx_path]
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
        """Join the background write of the last selection pickle (called before anything that reads or rewrites the files); an
        exception the write raised (disk full, missing directory, pickling error) is raised HERE -- where the reference's
        synchronous ``pickle.dump`` (:67-68) would have raised it one call earlier."""
        w, self._writer = self._writer, None
        if w is not None:
            w.join()

    @staticmethod
    def _writes_files():
        """Every rank of a data-parallel run holds the same prefix and lists: rank 0 writes them."""
        try:
            import torch.distributed as dist
            return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        except ImportError:
            return True

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
        if over.size and self._writes_files():
            fname = os.path.join(self.args.model_save_dir, '%s_selection_%02d.pkl' % (selection_method, self.selection_iter))
            self._writer = _Writer(fname, lambda sr=sr, n=n: sr.tuples(n))
        psup, lsup = pool.suppix, label.suppix
        plazy, llazy = isinstance(psup, LazySuppix), isinstance(lsup, LazySuppix)
        # Pool lists that exist as Python lists are edited in place, as the reference does: where every leaving id sits in its list,
        # for all entries at once -- in a list that holds its ids in ascending order (how the reference builds them, np.unique;
        # removals keep it) that is the number of listed ids below it, a row-wise cumsum of the touched rows of the table as it was
        # BEFORE this call; the positions are verified per picture before anything is deleted.  Lists nobody has asked for since the
        # table was built (LazySuppix) do not exist: their rows of the table are all there is to edit.
        order = np.argsort(img, kind='stable')              # entries of one picture together, in walk order
        simg, sids = img[order], ids[order]
        starts = np.concatenate(([0], np.flatnonzero(np.diff(simg)) + 1))
        ends = np.concatenate((starts[1:], [n]))
        first_seen = order[starts]                          # (stable sort: the first entry of a group is its first appearance)
        gkeys = [sr.keys[p] for p in simg[starts].tolist()]
        if plazy:
            real = np.fromiter((not isinstance(psup._raw(k[2]), _FromTable) for k in gkeys), dtype=bool, count=len(gkeys))
        else:
            real = np.ones(len(gkeys), dtype=bool)
        spos = None
        if real.any():
            grow = row_of[simg[starts]]                     # table row of every group
            need = np.flatnonzero(real)
            sub = np.cumsum(self._valid[grow[need]], axis=1, dtype=np.int16 if S < 32768 else np.int32)      # (before this call's removals: the table is updated below)
            slot = np.full(len(gkeys), -1, dtype=np.intp)
            slot[need] = np.arange(need.size)
            gidx = np.repeat(np.arange(len(gkeys)), ends - starts)
            spos = np.where(slot[gidx] >= 0, sub[np.maximum(slot[gidx], 0), sids].astype(np.intp) - 1, -1)
        self._valid[rows, ids] = 0
        if hasattr(pool, 'isselected'):
            pool.isselected[rows, ids] = 1
        left = self._valid[row_of[simg[starts]]].sum(axis=1, dtype=np.int64)      # ids that stay listed, per touched picture
        listed = {tuple(k) for k in label.im_idx}
        emptied = set()
        starts_l, ends_l, left_l, real_l = starts.tolist(), ends.tolist(), left.tolist(), real.tolist()
        for j in np.argsort(first_seen).tolist():
            a, b = starts_l[j], ends_l[j]
            key = gkeys[j]
            spx_path = key[2]
            if tuple(key) not in listed:
                listed.add(tuple(key))
                label.im_idx.append(key)
                if llazy:
                    dict.__setitem__(lsup, spx_path, _Appended())
                else:
                    lsup[spx_path] = []
            if llazy:
                cur = lsup._raw(spx_path)
                if isinstance(cur, _Appended):
                    cur.runs.append(sids[a:b])
                else:
                    cur.extend(sids[a:b].tolist())
            else:
                lsup[spx_path].extend(sids[a:b].tolist())
            if not real_l[j]:
                if left_l[j] == 0:
                    dict.__delitem__(psup, spx_path)
                    emptied.add(tuple(key))
                continue
            lst = psup._raw(spx_path) if plazy else psup[spx_path]
            if b - a < len(lst):
                sel = sids[a:b].tolist()
                ps = spos[a:b].tolist()
                m = len(lst)
                if all(q < m and lst[q] == i for q, i in zip(ps, sel)):     # verified positions: deleting them is right whatever the rest
                    for q in sorted(ps, reverse=True):
                        del lst[q]
                else:                                                        # a list in another order: rewrite it, order preserved
                    gone = set(sel)
                    psup[spx_path] = [i for i in lst if i not in gone]
            else:
                psup.pop(spx_path)
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
        if not self._writes_files():        # every rank of a data-parallel run holds the same lists: rank 0 writes them (and only it builds the lazy ones)
            return
        path = os.path.join(self.args.model_save_dir, 'datalist_%02d.pkl' % self.selection_iter)
        _atomic_pickle(path, {'trg_label_im_idx': self.trg_label_dataset.im_idx,
                              'trg_pool_im_idx': self.trg_pool_dataset.im_idx,
                              'trg_label_suppix': self.trg_label_dataset.suppix,
                              'trg_pool_suppix': self.trg_pool_dataset.suppix})

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
