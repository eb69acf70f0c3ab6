"""ResidentRegionDataset + RegionActiveDataset bookkeeping on the CPU (no kernels are launched: the datasets only
hold references to the pictures until a sample is drawn)."""
import types

import numpy as np
import torch

from mulactseg_amd.dataloader import RegionActiveDataset
from mulactseg_amd.dataloader.resident import ResidentRegionDataset


def _sets(tmp_path, n=3, nseg=16, C=20):
    args = types.SimpleNamespace(nseg=nseg, ignore_idx=255, fair_counting=True, or_labeling=True, model_save_dir=str(tmp_path),
                                 finetune_itrs=1)
    pics = [torch.zeros((8, 8, 3), dtype=torch.uint8) for _ in range(n)]
    spxs = [torch.zeros((8, 8), dtype=torch.int64) for _ in range(n)]
    rs = np.random.RandomState(0)
    mh = (rs.rand(n, nseg, C) < 0.1).astype(np.uint8)
    mh[..., 0] = 1
    names = [("im/a_%d.png" % k, "lb/a_%d.png" % k, "sp/a_%d.pkl" % k) for k in range(n)]
    start = {names[0][2]: [0, 1]}                                   # image 0 starts with two labelled regions
    pool_dict = {nm[2]: [i for i in range(nseg) if not (k == 0 and i in (0, 1))] for k, nm in enumerate(names)}
    label = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-label', region_dict=start)
    pool = ResidentRegionDataset(args, pics, spxs, mh, names, split='active-ulabel', region_dict=pool_dict)
    return args, names, mh, RegionActiveDataset(args, pool, label)


def test_expand_training_set_accumulates_on_an_already_labelled_image(tmp_path):
    """entries of im_idx are lists, so a second selection from a labelled image appends to its id list
    instead of overwriting it, no duplicate image entry appears, and an emptied pool image is removed."""
    args, names, mh, aset = _sets(tmp_path)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    assert label.im_idx == [list(names[0])] and label.suppix[names[0][2]] == [0, 1]
    order = [(0.9, ','.join(names[0]), 5), (0.8, ','.join(names[1]), 3), (0.7, ','.join(names[0]), 7), (0.6, ','.join(names[1]), 4)]
    n = aset.expand_training_set(order, 10 ** 6, 'x')
    assert n == 4
    assert label.suppix[names[0][2]] == [0, 1, 5, 7]
    assert label.suppix[names[1][2]] == [3, 4]
    assert label.im_idx == [list(names[0]), list(names[1])]          # no duplicates, lists as in the reference
    assert 5 not in pool.suppix[names[0][2]] and 7 not in pool.suppix[names[0][2]]
    assert pool.isselected[0, 5] == 1 and pool.isselected[1, 4] == 1 and pool.isselected.sum() == 4
    # empty one pool image completely: its key and its id list disappear
    rest = [(0.5, ','.join(names[2]), i) for i in range(args.nseg)]
    aset.expand_training_set(rest, 10 ** 6, 'x')
    assert names[2][2] not in pool.suppix and list(names[2]) not in pool.im_idx
    assert len(label) == 3 and len(pool) == 2


def test_budget_is_the_click_cost_under_fair_counting(tmp_path):
    args, names, mh, aset = _sets(tmp_path)
    order = [(1.0 - 0.01 * i, ','.join(names[1]), i) for i in range(args.nseg)]
    costs = np.cumsum([int(mh[1, i].sum()) for i in range(args.nseg)])
    budget = int(costs[4])                                            # reached exactly by 5 regions -> one more is taken
    n = aset.expand_training_set(order, budget, 'x')
    assert n == int(np.searchsorted(costs, budget, side='right')) + 1


def _reference_loop(pool, label, isselected, index_of, mh, sample_region, selection_count):
    """The loop of the reference's RegionActiveDataset.expand_training_set (dataloader/region_active_dataset.py:31-73,
    fair counting + or-labeling), statement by statement, on plain lists: the yardstick for the deferred-removal
    implementation.  Returns the number of consumed regions."""
    cost = 0
    for idx, (_, joined, sid) in enumerate(sample_region):
        key = joined.split(",")
        spx = key[2]
        if key not in label['im_idx']:
            label['im_idx'].append(key)
            label['suppix'][spx] = [sid]
        else:
            label['suppix'][spx].append(sid)
        pool['suppix'][spx].remove(sid)
        if len(pool['suppix'][spx]) == 0:
            pool['suppix'].pop(spx)
            pool['im_idx'].remove(key)
        isselected[index_of(spx), sid] = 1
        cost += int(mh[index_of(spx), sid].sum())
        if cost > selection_count:
            return idx + 1
    return len(sample_region)


def test_expand_training_set_equals_the_reference_loop_on_random_rounds(tmp_path):
    """Three successive rounds of random orders (one of them emptying pictures, one stopped by the budget) leave the lists,
    the id order inside every list, isselected, the valid table and the pickled prefix exactly as the reference loop does."""
    import copy
    import pickle
    rs = np.random.RandomState(4)
    args, names, mh, aset = _sets(tmp_path, n=6, nseg=24)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    ref_pool = {'im_idx': copy.deepcopy(pool.im_idx), 'suppix': copy.deepcopy(pool.suppix)}
    ref_label = {'im_idx': copy.deepcopy(label.im_idx), 'suppix': copy.deepcopy(label.suppix)}
    ref_sel = np.zeros_like(pool.isselected)
    index_of = lambda spx: label.id_to_index[spx.split('/')[-1].split('.')[0]]
    assert np.array_equal(aset.pool_valid_mask(args.nseg).sum(1), [len(pool.suppix[k[2]]) for k in pool.im_idx])
    for rnd, budget in enumerate([25, 10 ** 6, 40]):
        aset.selection_iter = rnd + 1
        cand = [(float(rs.rand()), ','.join(k), i) for k in pool.im_idx for i in pool.suppix[k[2]]]
        cand.sort(reverse=True)
        if rnd == 1:                                    # consume two whole pictures and a bit more, in score order
            whole = {','.join(pool.im_idx[0]), ','.join(pool.im_idx[2])}
            cand = [c for c in cand if c[1] in whole] + [c for c in cand if c[1] not in whole][:7]
        n_ref = _reference_loop(ref_pool, ref_label, ref_sel, index_of, mh, cand, budget)
        n = aset.expand_training_set(cand, budget, 'm')
        assert n == n_ref
        assert pool.im_idx == ref_pool['im_idx'] and pool.suppix == ref_pool['suppix']
        assert list(pool.suppix) == list(ref_pool['suppix'])                   # dict order too (datalist pickle)
        assert label.im_idx == ref_label['im_idx'] and label.suppix == ref_label['suppix']
        assert np.array_equal(pool.isselected, ref_sel)
        valid = aset.pool_valid_mask(args.nseg)
        for k, key in enumerate(pool.im_idx):
            assert sorted(np.nonzero(valid[k])[0].tolist()) == sorted(pool.suppix[key[2]])
        if n < len(cand):
            with open(tmp_path / ('m_selection_%02d.pkl' % (rnd + 1)), 'rb') as f:
                assert pickle.load(f) == cand[:n]
    # a region that is not in the pool any more raises as the reference does (list.remove -> ValueError; picture gone -> KeyError)
    gone = ref_label['suppix'][ref_label['im_idx'][0][2]][0]
    import pytest
    with pytest.raises((ValueError, KeyError)):
        aset.expand_training_set([(1.0, ','.join(ref_label['im_idx'][0]), gone)], 5, 'm')


def test_consumed_prefix_takes_the_array_path_and_equals_the_reference_loop(tmp_path):
    """What the device selection hands over -- a ConsumedPrefix (arrays that read as the reference's tuple list) -- goes through the
    grouped-by-picture path of expand_training_set: same lists, same order inside every list, same isselected / valid table and the
    same pickled prefix as the reference loop over the equivalent tuples, on three successive rounds (one stopped by the budget, one
    emptying pictures); an entry the reference would raise on falls back to the tuple walk and raises there."""
    import copy
    import pickle
    import pytest
    from mulactseg_amd.dataloader.region_active_dataset import ConsumedPrefix
    rs = np.random.RandomState(9)
    args, names, mh, aset = _sets(tmp_path, n=7, nseg=40)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    ref_pool = {'im_idx': copy.deepcopy(pool.im_idx), 'suppix': copy.deepcopy(pool.suppix)}
    ref_label = {'im_idx': copy.deepcopy(label.im_idx), 'suppix': copy.deepcopy(label.suppix)}
    ref_sel = np.zeros_like(pool.isselected)
    index_of = lambda spx: label.id_to_index[spx.split('/')[-1].split('.')[0]]
    for rnd, budget in enumerate([60, 10 ** 6, 45]):
        aset.selection_iter = rnd + 1
        aset.pool_valid_mask(args.nseg)                                 # (the selector builds the table before it selects)
        keys = list(pool.im_idx)
        cand = [(float(np.float32(rs.rand())), p, i) for p, k in enumerate(keys) for i in pool.suppix[k[2]]]
        cand.sort(key=lambda t: (-t[0], t[1], t[2]))
        if rnd == 1:                                                    # two whole pictures and a bit more
            cand = [c for c in cand if c[1] in (0, 3)] + [c for c in cand if c[1] not in (0, 3)][:9]
        sr = ConsumedPrefix(np.array([c[0] for c in cand], dtype=np.float32), [c[1] for c in cand], [c[2] for c in cand], keys)
        as_tuples = [(c[0], ','.join(keys[c[1]]), c[2]) for c in cand]
        assert len(sr) == len(cand) and sr[3] == as_tuples[3] and sr[:5] == as_tuples[:5] and list(sr)[-1] == as_tuples[-1]
        n_ref = _reference_loop(ref_pool, ref_label, ref_sel, index_of, mh, as_tuples, budget)
        n = aset.expand_training_set(sr, budget, 'm')
        aset.wait_for_writes()
        assert n == n_ref
        assert pool.im_idx == ref_pool['im_idx'] and pool.suppix == ref_pool['suppix'] and list(pool.suppix) == list(ref_pool['suppix'])
        assert label.im_idx == ref_label['im_idx'] and label.suppix == ref_label['suppix'] and list(label.suppix) == list(ref_label['suppix'])
        assert np.array_equal(pool.isselected, ref_sel)
        valid = aset.pool_valid_mask(args.nseg)
        for k, key in enumerate(pool.im_idx):
            assert sorted(np.nonzero(valid[k])[0].tolist()) == sorted(pool.suppix[key[2]])
        if n < len(cand):
            with open(tmp_path / ('m_selection_%02d.pkl' % (rnd + 1)), 'rb') as f:
                assert pickle.load(f) == as_tuples[:n]
    # an id that already left the pool: the array path declines, the tuple walk raises as list.remove does
    aset.pool_valid_mask(args.nseg)
    keys = list(pool.im_idx)
    gone_key = ref_label['im_idx'][-1]
    gone_id = ref_label['suppix'][gone_key[2]][0]
    if gone_key in keys:
        with pytest.raises((ValueError, KeyError)):
            aset.expand_training_set(ConsumedPrefix(np.ones(1, np.float32), [keys.index(gone_key)], [gone_id], keys), 5, 'm')


def test_removal_by_verified_position_and_its_fallback(tmp_path):
    """Ids leave a pool list by position (read off the valid table) when the list is ascending; a list in another order fails the
    position check and is rewritten instead -- both must equal the reference's list.remove() result, order included."""
    import copy
    args, names, mh, aset = _sets(tmp_path, n=2, nseg=40)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    rs = np.random.RandomState(1)
    shuffled = list(pool.suppix[names[1][2]])
    rs.shuffle(shuffled)
    pool.suppix[names[1][2]] = list(shuffled)                       # image 1: not ascending
    aset.pool_valid_mask(args.nseg)                                 # the table exists before the round
    want0 = [i for i in pool.suppix[names[0][2]] if i not in (3, 9, 10, 17, 30, 39)]
    want1 = [i for i in shuffled if i not in (2, 5, 8, 13, 21, 34, 0)]
    order = [(1.0 - 0.01 * k, ','.join(names[0]), i) for k, i in enumerate((30, 3, 39, 10, 9, 17))]
    order += [(0.5 - 0.01 * k, ','.join(names[1]), i) for k, i in enumerate((21, 2, 34, 8, 5, 13, 0))]
    calls = []
    orig = aset._delete_by_position
    aset._delete_by_position = lambda lst, row, gone: calls.append(orig(lst, row, gone)) or calls[-1]
    assert aset.expand_training_set(order, 10 ** 6, 'p') == len(order)
    assert calls == [True, False]
    assert pool.suppix[names[0][2]] == want0 and pool.suppix[names[1][2]] == want1
    assert label.suppix[names[0][2]][-6:] == [30, 3, 39, 10, 9, 17]


def test_valid_table_is_rebuilt_from_the_lists_when_a_round_ran_before_it_existed(tmp_path):
    """A pool may offer initial_valid_table() (all ids listed when it was built); once expand_training_set has changed the lists
    without a table to mirror it (a random first round), the table must come from the lists, not from the stale initial one."""
    args, names, mh, aset = _sets(tmp_path, n=3, nseg=16)
    pool = aset.trg_pool_dataset
    pool.initial_valid_table = lambda: np.ones((3, 16), dtype=np.uint8)
    aset.expand_training_set([(0.9, ','.join(names[1]), 3), (0.8, ','.join(names[2]), 5)], 10 ** 6, 'x')
    valid = aset.pool_valid_mask(16)
    for k, key in enumerate(pool.im_idx):
        assert sorted(np.nonzero(valid[k])[0].tolist()) == sorted(pool.suppix[key[2]])
    aset.expand_training_set([(0.7, ','.join(names[1]), 4)], 10 ** 6, 'x')          # now mirrored incrementally
    valid = aset.pool_valid_mask(16)
    assert valid[[k[2] for k in pool.im_idx].index(names[1][2]), 4] == 0


def test_lists_nobody_reads_stay_arrays_and_still_equal_the_reference_loop(tmp_path):
    """with the valid table in place and every pool list ascending, pool.suppix / label.suppix become LazySuppix
    mappings -- the array path of expand_training_set edits the table and appends id runs, Python lists appear when somebody reads
    them.  Four successive rounds WITHOUT a single list access in between (one round empties pictures, one picture is read half way
    and is edited as a real list from then on), then everything is compared with the reference loop: lists, order inside every list,
    dictionary order, im_idx, the datalist pickle (a plain dict of lists)."""
    import copy
    import pickle
    from mulactseg_amd.dataloader.region_active_dataset import ConsumedPrefix, LazySuppix
    rs = np.random.RandomState(21)
    args, names, mh, aset = _sets(tmp_path, n=9, nseg=48)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    ref_pool = {'im_idx': copy.deepcopy(pool.im_idx), 'suppix': copy.deepcopy(pool.suppix)}
    ref_label = {'im_idx': copy.deepcopy(label.im_idx), 'suppix': copy.deepcopy(label.suppix)}
    ref_sel = np.zeros_like(pool.isselected)
    index_of = lambda spx: label.id_to_index[spx.split('/')[-1].split('.')[0]]
    aset.pool_valid_mask(args.nseg)
    assert isinstance(pool.suppix, LazySuppix) and isinstance(label.suppix, LazySuppix)
    assert pool.suppix.pending() == 9 and len(pool.suppix) == 9 and names[3][2] in pool.suppix
    for rnd, budget in enumerate([70, 10 ** 6, 55, 10 ** 6]):
        aset.selection_iter = rnd + 1
        valid = aset.pool_valid_mask(args.nseg)                         # candidates from the TABLE (what the selector reads)
        keys = list(pool.im_idx)
        cand = [(float(np.float32(rs.rand())), p, int(i)) for p in range(len(keys)) for i in np.flatnonzero(valid[p])]
        cand.sort(key=lambda t: (-t[0], t[1], t[2]))
        if rnd == 1:                                                    # two whole pictures and a bit more
            cand = [c for c in cand if c[1] in (1, 4)] + [c for c in cand if c[1] not in (1, 4)][:11]
        if rnd == 3:
            cand = cand[:40]
        sr = ConsumedPrefix(np.array([c[0] for c in cand], dtype=np.float32), [c[1] for c in cand], [c[2] for c in cand], keys)
        as_tuples = [(c[0], ','.join(keys[c[1]]), c[2]) for c in cand]
        n_ref = _reference_loop(ref_pool, ref_label, ref_sel, index_of, mh, as_tuples, budget)
        before = pool.suppix.pending()
        assert aset.expand_training_set(sr, budget, 'm') == n_ref
        assert pool.suppix.pending() in (before, before - 2)            # (round 1 drops two emptied pictures; nothing was built)
        if rnd == 2:                                                    # somebody reads one pool list and one label list ...
            k5 = pool.im_idx[5][2]
            assert pool.suppix[k5] == ref_pool['suppix'][k5] and label.suppix[k5] == ref_label['suppix'][k5]
            assert isinstance(pool.suppix[k5], list) and pool.suppix[k5] is pool.suppix[k5]     # ... they are real lists from now on
    aset.wait_for_writes()
    assert pool.suppix.pending() >= 5
    assert pool.im_idx == ref_pool['im_idx'] and label.im_idx == ref_label['im_idx']
    assert list(pool.suppix) == list(ref_pool['suppix']) and list(label.suppix) == list(ref_label['suppix'])
    assert np.array_equal(pool.isselected, ref_sel)
    aset.dump_datalist()
    with open(tmp_path / 'datalist_04.pkl', 'rb') as f:
        data = pickle.load(f)
    assert type(data['trg_pool_suppix']) is dict and type(data['trg_label_suppix']) is dict
    assert data['trg_pool_suppix'] == ref_pool['suppix'] and data['trg_label_suppix'] == ref_label['suppix']
    assert list(data['trg_pool_suppix']) == list(ref_pool['suppix'])
    assert pool.suppix == ref_pool['suppix'] and label.suppix == ref_label['suppix'] and pool.suppix.pending() == 0
    # the tuple path on the same mappings (a selector without calculate_scores_tensor) keeps working
    key = pool.im_idx[0]
    sid = pool.suppix[key[2]][0]
    _reference_loop(ref_pool, ref_label, ref_sel, index_of, mh, [(1.0, ','.join(key), sid)], 5)
    aset.expand_training_set([(1.0, ','.join(key), sid)], 5, 'm')
    assert pool.suppix == ref_pool['suppix'] and label.suppix == ref_label['suppix']


def test_tuple_path_on_lists_nobody_has_built_equals_the_reference_loop(tmp_path):
    """the tuple path of expand_training_set on pool lists LazySuppix has NOT built (pending() > 0).  Removing two
    ids used to raise ValueError after half the state had been edited; removing 10 of 16 ids silently dropped the picture."""
    import copy
    from mulactseg_amd.dataloader.region_active_dataset import LazySuppix
    args, names, mh, aset = _sets(tmp_path)
    pool, label = aset.trg_pool_dataset, aset.trg_label_dataset
    ref_pool = {'im_idx': copy.deepcopy(pool.im_idx), 'suppix': copy.deepcopy(pool.suppix)}
    ref_label = {'im_idx': copy.deepcopy(label.im_idx), 'suppix': copy.deepcopy(label.suppix)}
    ref_sel = np.zeros_like(pool.isselected)
    index_of = lambda spx: label.id_to_index[spx.split('/')[-1].split('.')[0]]
    aset.pool_valid_mask(args.nseg)
    assert isinstance(pool.suppix, LazySuppix) and pool.suppix.pending() == 3
    two = [(0.9, ','.join(names[1]), 3), (0.8, ','.join(names[1]), 9)]
    ten = [(0.7 - 0.01 * i, ','.join(names[2]), i) for i in range(10)]
    whole = [(0.5 - 0.01 * i, ','.join(names[0]), i) for i in range(2, args.nseg)]          # (0 and 1 are labelled from the start)
    for rnd, order in enumerate([two, ten, whole]):
        aset.selection_iter = rnd + 1
        n_ref = _reference_loop(ref_pool, ref_label, ref_sel, index_of, mh, order, 10 ** 6)
        assert aset.expand_training_set(order, 10 ** 6, 't') == n_ref
        assert pool.suppix.pending() > 0                                # still nothing of the pool was built by the call itself
        assert pool.im_idx == ref_pool['im_idx'] and list(pool.suppix) == list(ref_pool['suppix'])
    assert names[0][2] not in pool.suppix and names[2][2] in pool.suppix
    assert pool.suppix == ref_pool['suppix'] and label.suppix == ref_label['suppix']
    assert pool.im_idx == ref_pool['im_idx'] and label.im_idx == ref_label['im_idx'] and np.array_equal(pool.isselected, ref_sel)
    try:                                                                # an id that already left: raises BEFORE anything is edited
        aset.expand_training_set([(1.0, ','.join(names[1]), 3)], 5, 't')
        assert False
    except ValueError:
        pass
    assert pool.suppix == ref_pool['suppix'] and label.suppix == ref_label['suppix']


def test_lazy_suppix_reads_like_a_dict_of_lists():
    import copy
    import pickle
    from mulactseg_amd.dataloader.region_active_dataset import LazySuppix, _Appended, _FromTable
    tab = np.array([[1, 0, 1, 1], [0, 0, 0, 1]], dtype=np.uint8)
    d = LazySuppix()
    dict.__setitem__(d, 'a', _FromTable(tab, 0))
    dict.__setitem__(d, 'b', _FromTable(tab, 1))
    ap = _Appended([7])
    ap.runs += [np.array([3, 1]), np.array([9])]
    dict.__setitem__(d, 'c', ap)
    assert len(d) == 3 and 'a' in d and list(d) == ['a', 'b', 'c'] and d.pending() == 3
    tab[0, 0] = 0                                        # the table is edited in place until the list is asked for
    assert d['a'] == [2, 3] and d.pending() == 2 and d.get('zz', 5) == 5 and d.get('b') == [3]
    assert d['c'] == [7, 3, 1, 9]
    d['a'].append(11)                                    # a handed-out list is the list
    assert d['a'] == [2, 3, 11]
    assert dict(d.items()) == {'a': [2, 3, 11], 'b': [3], 'c': [7, 3, 1, 9]} and sorted(map(len, d.values())) == [1, 3, 4]
    e = pickle.loads(pickle.dumps(d))
    assert type(e) is dict and e == d and d == e and copy.deepcopy(d) == e and not (d != e)
    assert d.pop('b') == [3] and 'b' not in d and d.pop('b', None) is None
    assert d.setdefault('q', []) == [] and d.setdefault('a') == [2, 3, 11]
    assert 'FromTable' not in repr(d)
    assert dict(d) == {'a': [2, 3, 11], 'c': [7, 3, 1, 9], 'q': []} and {**d}['c'] == [7, 3, 1, 9]
    f = LazySuppix()
    dict.__setitem__(f, 'z', _FromTable(tab, 1))
    plain = {}
    plain.update(f)
    assert plain == {'z': [3]} and type(dict(f)['z']) is list


def test_a_failed_selection_pickle_write_is_raised_not_lost(tmp_path):
    """the background write of <method>_selection_RR.pkl must not fail silently -- wait_for_writes (called by the next
    expand / dump / load) re-raises what the thread raised, and no temporary file is left behind."""
    import os
    import pytest
    from mulactseg_amd.dataloader.region_active_dataset import ConsumedPrefix
    args, names, mh, aset = _sets(tmp_path, n=3, nseg=16)
    aset.pool_valid_mask(args.nseg)
    args.model_save_dir = str(tmp_path / "does" / "not" / "exist")
    keys = list(aset.trg_pool_dataset.im_idx)
    sr = ConsumedPrefix(np.linspace(1, 0.5, 6).astype(np.float32), [1] * 6, [2, 3, 4, 5, 6, 7], keys)
    assert aset.expand_training_set(sr, 3, 'm') < 6                      # stopped by the budget: a prefix is written
    with pytest.raises(OSError):
        aset.wait_for_writes()
    aset.wait_for_writes()                                               # (reported once)
    assert not [f for f in os.listdir(tmp_path) if '.tmp.' in f]
