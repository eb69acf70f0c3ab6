"""Active-set bookkeeping with the semantics of the reference's
``dataloader/region_active_dataset.py:8-105`` (same public methods and pickle layouts).

``expand_training_set`` moves regions from the pool to the labelled set in the order given and stops
after the region that makes the click cost exceed the budget; the consumed prefix is pickled as
``<method>_selection_RR.pkl`` and the lists as ``datalist_RR.pkl``.
"""
import os
import pickle


class RegionActiveDataset:
    def __init__(self, args, trg_pool_dataset, trg_label_dataset):
        self.args = args
        self.selection_iter = 0
        self.trg_pool_dataset = trg_pool_dataset
        self.trg_label_dataset = trg_label_dataset

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

    # -- selection ------------------------------------------------------------------------------
    def expand_training_set(self, sample_region, selection_count, selection_method):
        """``sample_region``: sorted list of (score, "img,lbl,spx", suppix_id)."""
        pool, label = self.trg_pool_dataset, self.trg_label_dataset
        cost = 0
        n_sup = 0
        # ``key not in label.im_idx`` of the reference (:38) is a linear scan per region (2 975 list compares x 100 000
        # regions per Cityscapes round); the set below answers the same question: an image is listed iff its key is.
        listed = {tuple(k) for k in label.im_idx}
        for idx, (_, joined, suppix_id) in enumerate(sample_region):
            key = joined.split(",")
            spx_path = key[2]
            if tuple(key) not in listed:
                listed.add(tuple(key))
                label.im_idx.append(key)
                label.suppix[spx_path] = [suppix_id]
            else:
                label.suppix[spx_path].append(suppix_id)
            pool.suppix[spx_path].remove(suppix_id)
            if len(pool.suppix[spx_path]) == 0:
                pool.suppix.pop(spx_path)
                pool.im_idx.remove(key)
            if hasattr(pool, 'isselected'):
                pool.isselected[self._image_index(spx_path), suppix_id] = 1
            cost += self.region_cost(spx_path, suppix_id)
            n_sup += 1
            if cost > selection_count:
                fname = '%s_selection_%02d.pkl' % (selection_method, self.selection_iter)
                with open(os.path.join(self.args.model_save_dir, fname), "wb") as f:
                    pickle.dump(sample_region[:idx + 1], f)
                break
        log = getattr(getattr(self.args, 'wandb', None), 'log', None)
        if log is not None and n_sup:
            step = int(getattr(self.args, 'finetune_itrs', 0)) * (self.selection_iter - 1)
            log({"num_selected_spx": n_sup, "num_cls_spx": selection_count / n_sup,
                 "sampling_iter": self.selection_iter}, step=step)
        return n_sup

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

    def get_trainset(self):
        return self.trg_label_dataset
