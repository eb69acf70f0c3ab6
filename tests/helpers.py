"""Test helpers: an oracle-backed stand-in for the HIP backend (CPU tests of the host logic only) and
tiny fake trainer / pool objects shaped like the reference's plugin arguments."""
import types

import numpy as np
import torch

from oracle import exact, port


class OracleBackend:
    """Implements the backend protocol of mulactseg_amd.active_selection.engine with the CPU oracle.
    TEST INFRASTRUCTURE: lives under tests/, is never importable from the product package."""
    name = "oracle"

    def __init__(self):
        self.device = torch.device('cpu')

    def inv_temperature(self, T):
        return float(exact.inv_temperature(T))

    def class_prob_sum(self, logits, invT, out):
        out += torch.from_numpy(exact.class_prob_sum(logits.numpy(), np.float32(invT)).view(np.int64))

    def region_accum(self, logits, spx, cls_w, S, invT, score_sum, hist):
        s, h = exact.bvsb_region_accum(logits.numpy(), spx.numpy(), None if cls_w is None else cls_w.numpy(), S,
                                       np.float32(invT))
        score_sum += torch.from_numpy(s.view(np.int64))
        hist += torch.from_numpy(h.view(np.int32))

    def finalize(self, score_sum, hist, ban_class, want_hist_i64=False):
        score, dom, cnt = exact.region_finalize(score_sum.numpy().view(np.uint64), hist.numpy().view(np.uint32), ban_class)
        h64 = hist.to(torch.int64) if want_hist_i64 else None
        return torch.from_numpy(score), torch.from_numpy(dom), torch.from_numpy(cnt.view(np.int32)), h64

    def single_pass(self, logits, spx, S, invT, prob_sum, class_sum, hist):
        ps, cs, h = exact.single_pass_accum(logits.numpy(), spx.numpy(), S, np.float32(invT))
        prob_sum += torch.from_numpy(ps.view(np.int64))
        class_sum += torch.from_numpy(cs.view(np.int64))
        hist += torch.from_numpy(h.view(np.int32))

    def single_pass_lowres(self, zq, size, spx, S, invT, prob_sum, class_sum, hist):
        full = exact.upsample_bilinear(zq.numpy(), int(size[0]), int(size[1]))
        self.single_pass(torch.from_numpy(full), spx, S, invT, prob_sum, class_sum, hist)

    def class_weight(self, prob_sum, hw, batch_size, n_batches, coeff):
        from mulactseg_amd.active_selection.engine import class_weight_from_sums
        n_img = prob_sum.shape[0]
        cum, w = class_weight_from_sums(prob_sum.numpy(), hw, np.arange(n_img) // batch_size, n_batches, coeff)
        return cum, torch.from_numpy(w)

    def finalize_weighted(self, class_sum, hist, cls_w, ban_class, want_hist_i64=False):
        w = np.ones(hist.shape[-1], dtype=np.float32) if cls_w is None else cls_w.numpy()
        score, dom, cnt = exact.region_finalize_weighted(class_sum.numpy().view(np.uint64), hist.numpy().view(np.uint32),
                                                         exact.weights_to_fixed31(w), ban_class)
        h64 = hist.to(torch.int64) if want_hist_i64 else None
        return torch.from_numpy(score), torch.from_numpy(dom), torch.from_numpy(cnt.view(np.int32)), h64

    def minmax_normalize_(self, scores):
        u = scores.numpy()
        mn = u[u != 0].min()
        u -= mn
        u /= u.max()
        return scores

    def region_reweight_(self, scores, dominant, ban_class, cls_w):
        s, d = scores.numpy(), dominant.numpy()
        s[d == ban_class] = 0
        if cls_w is not None:
            s *= cls_w.numpy()[d]
        return scores

    def dominant_hist(self, dominant, C):
        return torch.from_numpy(np.bincount(dominant.numpy().ravel(), minlength=C).astype(np.int64))

    def select(self, scores, valid, img_rank, img_of_rank, region_cost, budget, max_out):
        # reference semantics: Python tuple sort with the rank standing in for the path string
        s = scores.numpy()
        v = np.ones(s.shape, dtype=bool) if valid is None else valid.numpy()        # (None: every region is still in the pool)
        rank = img_rank.numpy()
        tuples = [(float(s[i, r]), int(rank[i]), r) for i in range(s.shape[0]) for r in range(s.shape[1]) if v[i, r]]
        inv = img_of_rank.numpy()
        cost = None if region_cost is None else region_cost.numpy()
        fn = None if cost is None else (lambda rk, rid: int(cost[inv[rk], rid]))
        taken = port.select_regions(tuples, budget, fn)
        return (len(taken), np.array([inv[t[1]] for t in taken], dtype=np.int32),
                np.array([t[2] for t in taken], dtype=np.int32), np.array([t[0] for t in taken], dtype=np.float32))


    def select_sharded(self, plan, scores, valid, img_rank, img_of_rank, region_cost, budget, max_out):
        """The sharded K4 of engine.select_regions restated on tuples: local head of max_out regions, all-gather, merged walk."""
        import torch.distributed as dist
        s = scores.numpy()
        v = np.ones(s.shape, dtype=bool) if valid is None else valid.numpy()
        rank = img_rank.numpy()
        local = sorted(((float(s[i, r]), int(rank[i]), r) for i in range(plan.img_lo, plan.img_hi) for r in range(s.shape[1]) if v[i, r]),
                       reverse=True)[:max_out]
        heads = [None] * plan.world
        dist.all_gather_object(heads, local)
        merged = sorted((t for h in heads for t in h), reverse=True)
        inv = img_of_rank.numpy()
        cost = None if region_cost is None else region_cost.numpy()
        fn = None if cost is None else (lambda rk, rid: int(cost[inv[rk], rid]))
        taken = port.select_regions(merged, budget, fn)
        return (len(taken), np.array([inv[t[1]] for t in taken], dtype=np.int32),
                np.array([t[2] for t in taken], dtype=np.int32), np.array([t[0] for t in taken], dtype=np.float32))


class FakePool(torch.utils.data.Dataset):
    """Pool dataset whose 'images' ARE the logits (the fake trainer's net is the identity)."""

    def __init__(self, logits, spx, im_idx, suppix):
        self.logits, self.spx = torch.from_numpy(logits), torch.from_numpy(spx)
        self.im_idx = [list(k) for k in im_idx]
        self.suppix = {k: list(v) for k, v in suppix.items()}

    def __len__(self):
        return len(self.im_idx)

    def __getitem__(self, i):
        return {'images': self.logits[i], 'spx': self.spx[i]}


def fake_trainer(device='cpu', save_dir=None):
    return types.SimpleNamespace(net=torch.nn.Identity(), device=torch.device(device), model_save_dir=save_dir,
                                 selection_iter=1)


def selector_args(**kw):
    base = dict(val_batch_size=2, val_num_workers=0, nseg=64, active_method='x', num_classes=19, ce_temp=0.1,
                cls_weight_coeff=6.0, method='active_joint_multi_predignore_lossdecomp', save_scores=False,
                fair_counting=True, or_labeling=True, model_save_dir=None, finetune_itrs=1,
                wandb=types.SimpleNamespace(log=lambda *a, **k: None))
    base.update(kw)
    return types.SimpleNamespace(**base)


class OracleLossOps:
    """CPU stand-in for the part of mulactseg_amd.ops the loss modules call (partial_loss_fwd_fused / _bwd_fused, LossState,
    inv_temperature), backed by oracle/exact.c -- lets the world-size-2 gloo tests drive FusedPartialLabelLoss's normaliser
    all-reduce and autograd wiring without a GPU.  Full-resolution logits only.  TEST INFRASTRUCTURE."""

    class LossState:
        pass

    @staticmethod
    def inv_temperature(T):
        return float(exact.inv_temperature(T))

    @staticmethod
    def partial_loss_fwd_fused(z, size, spx, mask, invT, flags, targets=None, cols_used=None, bits=None, weights=None, reduce_acc=None):
        import ctypes
        assert size is None and weights is None
        if bits is None:
            bits = torch.from_numpy(exact.target_bits(targets.numpy(), cols_used).view(np.int32))
        acc, gmax, _ = exact.partial_loss_fwd(z.detach().numpy(), spx.numpy(), mask.numpy().astype(np.uint8), bits.numpy().view(np.uint32),
                                              np.float32(invT), flags)
        acc_t = torch.from_numpy(acc.view(np.int64).copy())
        if reduce_acc is not None:
            reduce_acc(acc_t)                            # the all-reduce of the integer sums and counts under test
        accn = np.ascontiguousarray(acc_t.numpy().view(np.uint64))
        losses = np.zeros(3, dtype=np.float32)
        exact.lib().exact_loss_values(accn.ctypes.data_as(ctypes.c_void_p), flags, losses.ctypes.data_as(ctypes.c_void_p))
        st = OracleLossOps.LossState()
        st.acc, st.flags = acc_t, flags
        st.N, st.S, st.C = z.shape[0], bits.shape[1], z.shape[1]
        st.bits = bits
        # (what the product keeps in its work buffer: acc | table | bit masks)
        st.work = torch.cat([acc_t, torch.from_numpy(gmax.view(np.int64)).reshape(-1), bits.reshape(-1).to(torch.int64)])
        return torch.from_numpy(losses), st

    @staticmethod
    def partial_loss_bwd_fused(z, size, spx, mask, state, grad, invT, weights=None, want_fix=False):
        assert size is None and weights is None
        n_tab = state.N * state.S * state.C
        acc = state.work[:8]
        gmax = state.work[8:8 + n_tab].reshape(state.N, state.S, state.C)
        bits = state.work[8 + n_tab:].to(torch.int32).reshape(state.N, state.S)
        _, dz = exact.partial_loss_bwd(z.detach().numpy(), spx.numpy(), mask.numpy().astype(np.uint8), bits.numpy().view(np.uint32),
                                       gmax.numpy().view(np.uint64), acc.numpy().view(np.uint64), grad.numpy(), np.float32(invT), state.flags)
        return torch.from_numpy(dz)


# ---- test-support library (tests/libmulactseg_test.so, built from mulactseg_amd/csrc/test_support.hip by build()) --------------
_TEST_LIB = None


def _test_lib():
    """TEST INFRASTRUCTURE: the CU-hogging neighbour kernel lives in its own library, not in the product ABI."""
    global _TEST_LIB
    if _TEST_LIB is None:
        import ctypes
        import os
        from mulactseg_amd import _lib
        _lib.load()                                  # (torch's HIP runtime first: one runtime per process)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmulactseg_test.so")
        if not os.path.exists(path):
            raise RuntimeError("%s is not built: run `make -C mulactseg_amd/csrc`" % path)
        lib = ctypes.CDLL(path)
        lib.mas_test_occupy.restype = ctypes.c_int
        lib.mas_test_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p]
        _TEST_LIB = lib
    return _TEST_LIB


def occupy_cus(blocks, lds_bytes, seconds, stream=None):
    """`blocks` workgroups that hold 256 threads + `lds_bytes` of LDS for `seconds` (bounded: every wave leaves after that)."""
    from mulactseg_amd import _lib
    st = torch.cuda.current_stream() if stream is None else stream
    _lib.check(_test_lib().mas_test_occupy(int(blocks), int(lds_bytes), int(seconds * 1e8), st.cuda_stream), "mas_test_occupy")


# ---------------------------------------------------------------------------------------------------------------------------
# a tiny Cityscapes-shaped directory tree in the reference's on-disk formats
# ---------------------------------------------------------------------------------------------------------------------------
CITY_TRAIN_IDS = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]     # raw ids of the 19 training classes


def write_cityscapes_tree(root, n=5, H=128, W=256, nseg=64, n_val=2, seed=0, trim=5):
    """Writes pictures (``leftImg8bit/train/<city>/*_leftImg8bit.png``), raw label PNGs (``gtFine/train/<city>/*_gtFine_labelIds.png``),
    superpixel pickles (``superpixel_seed/cityscapes/seeds_<nseg>/train/label/*.pkl`` = ``{'labels': int32 [H,W], 'valid_idxes': ...}``),
    the multi-hot tensor + sizes (``.../gtFine_multi_tensor_trim_<k>x<k>/{multi_hot_cls,sp_size}.npy``), the target datalist
    (``train_seed<nseg>_or.txt``: picture, ``gtFine_or/<stem>.npy``, superpixel file -- tab separated), the region dictionary
    (``{spx path: [nseg, [missing ids]]}``) and a validation list (``val.txt``, whitespace separated, labelIds PNGs).
    Returns a dict of the paths and the arrays that were written."""
    import json
    import os
    import pickle
    from PIL import Image
    from mulactseg_amd import synth
    rs = np.random.RandomState(seed)
    cities = ['aachen', 'bochum']
    out = {'root': str(root), 'pictures': [], 'raw_labels': [], 'train_ids': [], 'spx': [], 'stems': [], 'H': H, 'W': W, 'nseg': nseg, 'val_lines': []}
    spx_dir = os.path.join(root, 'superpixel_seed/cityscapes/seeds_%d/train' % nseg)
    os.makedirs(os.path.join(spx_dir, 'label'), exist_ok=True)
    mt_dir = os.path.join(spx_dir, 'gtFine_multi_tensor_trim_%dx%d' % (trim, trim))
    os.makedirs(mt_dir, exist_ok=True)
    lines, region = [], {}
    multi_hot = np.zeros((n, nseg, 20), dtype=np.uint8)
    sizes = np.zeros((n, nseg), dtype=np.int32)
    lut = np.full(256, 255, dtype=np.int64)
    lut[CITY_TRAIN_IDS] = np.arange(19)
    for k in range(n + n_val):
        split = 'train' if k < n else 'val'
        city = cities[k % 2]
        stem = '%s_%06d_000019' % (city, k)
        cls = synth.class_map(seed * 1000 + k, H, W, 19, blob=16)                       # training ids 0..18
        raw = np.asarray(CITY_TRAIN_IDS, dtype=np.uint8)[cls]
        raw[rs.uniform(size=raw.shape) < 0.04] = 3                                       # some "out of roi" pixels -> ignored (255)
        pic = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        for sub in ('leftImg8bit', 'gtFine'):
            os.makedirs(os.path.join(root, sub, split, city), exist_ok=True)
        img_rel = 'leftImg8bit/%s/%s/%s_leftImg8bit.png' % (split, city, stem)
        lbl_rel = 'gtFine/%s/%s/%s_gtFine_labelIds.png' % (split, city, stem)
        Image.fromarray(pic).save(os.path.join(root, img_rel))
        Image.fromarray(raw).save(os.path.join(root, lbl_rel))
        if split == 'val':
            out['val_lines'].append('%s %s %s' % (img_rel, lbl_rel, lbl_rel))
            continue
        spx = synth.superpixel_map(seed * 1000 + 500 + k, H, W, nseg).astype(np.int32)
        spx_rel = 'superpixel_seed/cityscapes/seeds_%d/train/label/%s.pkl' % (nseg, stem)
        with open(os.path.join(root, spx_rel), 'wb') as f:
            pickle.dump({'labels': spx, 'valid_idxes': np.unique(spx)}, f)
        tid = lut[raw]
        col = np.where(tid == 255, 19, tid)
        multi_hot[k, spx.reshape(-1), col.reshape(-1)] = 1
        sizes[k] = np.bincount(spx.reshape(-1), minlength=nseg)[:nseg]
        missing = sorted(set(range(nseg)) - set(np.unique(spx).tolist()))
        region[spx_rel] = [nseg, missing]
        lines.append('\t'.join([img_rel, 'superpixel_seed/cityscapes/seeds_%d/train/gtFine_or/%s.npy' % (nseg, stem), spx_rel]))
        out['pictures'].append(pic), out['raw_labels'].append(raw), out['train_ids'].append(tid), out['spx'].append(spx), out['stems'].append(stem)
    np.save(os.path.join(mt_dir, 'multi_hot_cls.npy'), multi_hot)
    np.save(os.path.join(mt_dir, 'sp_size.npy'), sizes)
    lists = os.path.join(root, 'lists')
    os.makedirs(lists, exist_ok=True)
    out['trg_datalist'] = os.path.join(lists, 'train_seed%d_or.txt' % nseg)
    out['region_dict'] = os.path.join(lists, 'train_seed%d.dict' % nseg)
    out['val_datalist'] = os.path.join(lists, 'val.txt')
    with open(out['trg_datalist'], 'w') as f:
        f.write('\n'.join(lines) + '\n')
    with open(out['region_dict'], 'w') as f:
        json.dump(region, f)
    with open(out['val_datalist'], 'w') as f:
        f.write('\n'.join(out['val_lines']) + '\n')
    out['multi_hot'], out['sp_size'], out['lines'] = multi_hot, sizes, lines
    return out


def cityscapes_tree_args(tree, save_dir, extra=()):
    """The flags of the stage-1 production run (``script/open_source/train_city_mul_res50.sh``) pointed at ``tree``."""
    from mulactseg_amd.utils.common import get_parser
    a = get_parser().parse_args([
        '-m', 'deeplabv3pluswn_resnet50deepstem', '--separable_conv', '--method', 'active_joint_multi_predignore_lossdecomp',
        '--active_method', 'my_bvsb_predclsbal_pwr_banignore', '--cls_weight_coeff', '6.0', '--or_labeling', '--fair_counting',
        '--loss_type', 'joint_multi_loss', '--nseg', str(tree['nseg']), '--scheduler', 'poly', '--train_lr', '0.00002',
        '--train_transform', 'rescale_769_multi_notrg', '--loader', 'region_cityscapes_or_tensor', '--multi_ce_temp', '0.1',
        '--group_ce_temp', '0.1', '--ce_temp', '0.1', '--coeff', '16.0', '--coeff_mc', '8.0', '--coeff_gm', '1.0',
        '--trim_kernel_size', '5', '--trim_multihot_boundary', '--trg_data_dir', tree['root'], '--trg_datalist', tree['trg_datalist'],
        '--region_dict', tree['region_dict'], '--val_data_dir', tree['root'], '--val_datalist', tree['val_datalist'],
        '--train_batch_size', '2', '--val_batch_size', '2', '--num_workers', '0', '--val_num_workers', '0', '--finetune_itrs', '2',
        '--val_period', '2', '--log_period', '1', '--active_selection_size', '40', '-p', str(save_dir)] + list(extra))
    a.pretrained_backbone = False
    return a


def write_voc_tree(root, n=4, nseg=150, seed=0, trim=5, sizes=((120, 160), (150, 110), (96, 128), (130, 130))):
    """A tiny PASCAL-VOC-shaped tree in the reference's layout (dataloader/region_voc.py:75-84, region_voc_or_tensor.py:30-62):
    ``VOC2012/JPEGImages/<name>.jpg``, ``VOC2012/SegmentationClass/<name>.png`` (palette PNG of class indices, 255 = void),
    ``superpixels/pascal_voc_seg/seeds_32/train/label/<name>.pkl``, the multi-hot tensor with its "undefined" column
    (``.../gtFine_multi_tensor_trim_<k>x<k>/multi_hot_cls.npy`` u8 [n, nseg, 22]), a datalist of bare names and a region dictionary
    keyed by them.  Pictures have DIFFERENT sizes, as in VOC."""
    import json
    import os
    import pickle
    from PIL import Image
    from mulactseg_amd import synth
    rs = np.random.RandomState(seed)
    out = {'root': str(root), 'names': [], 'classes': [], 'spx': [], 'nseg': nseg}
    for d in ('VOC2012/JPEGImages', 'VOC2012/SegmentationClass', 'superpixels/pascal_voc_seg/seeds_32/train/label',
              'superpixels/pascal_voc_seg/seeds_32/train/gtFine_multi_tensor_trim_%dx%d' % (trim, trim), 'lists'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    mh = np.zeros((n, nseg, 22), dtype=np.uint8)
    region = {}
    for k in range(n):
        H, W = sizes[k % len(sizes)]
        name = '2007_%06d' % (32 + 7 * k)
        cls = synth.class_map(seed * 100 + k, H, W, 21, blob=12).astype(np.uint8)
        cls[rs.uniform(size=cls.shape) < 0.05] = 255
        n_ids = 40 + 10 * k
        spx = (synth.superpixel_map(seed * 100 + 50 + k, H, W, n_ids) % n_ids).astype(np.int32)
        Image.fromarray(rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)).save(os.path.join(root, 'VOC2012/JPEGImages', name + '.jpg'), quality=95)
        pal = Image.fromarray(cls, mode='P')
        pal.putpalette([v for i in range(256) for v in ((i * 37) % 256, (i * 91) % 256, (i * 13) % 256)])
        pal.save(os.path.join(root, 'VOC2012/SegmentationClass', name + '.png'))
        with open(os.path.join(root, 'superpixels/pascal_voc_seg/seeds_32/train/label', name + '.pkl'), 'wb') as f:
            pickle.dump({'labels': spx}, f)
        col = np.where(cls == 255, 21, cls)
        mh[k, spx.reshape(-1), col.reshape(-1)] = 1
        region[name] = [int(spx.max()) + 1, sorted(set(range(int(spx.max()) + 1)) - set(np.unique(spx).tolist()))]
        out['names'].append(name), out['classes'].append(cls), out['spx'].append(spx)
    np.save(os.path.join(root, 'superpixels/pascal_voc_seg/seeds_32/train/gtFine_multi_tensor_trim_%dx%d' % (trim, trim), 'multi_hot_cls.npy'), mh)
    out['trg_datalist'] = os.path.join(root, 'lists', 'train_seed%d_or.txt' % nseg)
    out['region_dict'] = os.path.join(root, 'lists', 'train_seed%d.dict' % nseg)
    out['val_datalist'] = os.path.join(root, 'lists', 'val.txt')
    for path, names in ((out['trg_datalist'], out['names']), (out['val_datalist'], out['names'][:2])):
        with open(path, 'w') as f:
            f.write('\n'.join(names) + '\n')
    with open(out['region_dict'], 'w') as f:
        json.dump(region, f)
    out['multi_hot'] = mh
    return out
