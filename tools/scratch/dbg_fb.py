import sys, os, hashlib, tempfile, pathlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch, numpy as np
import test_filebacked_gpu as T
import helpers

def h(t):
    return hashlib.md5(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:8]

from mulactseg_amd.trainer import active_joint_multi_predignore_lossdecomp as TR
orig_batch = TR.ActiveTrainer._batch
def _batch(self):
    out = orig_batch(self)
    print('   batch', [h(o) for o in out], [tuple(o.shape) for o in out], flush=True)
    return out
TR.ActiveTrainer._batch = _batch
orig_fwd = TR.ActiveTrainer.forward_train
def fwd(self, images, **kw):
    y = orig_fwd(self, images, **kw)
    print('   logits', h(y), 'params', h(torch.cat([p.detach().reshape(-1)[:4] for p in self.net.parameters()])), flush=True)
    return y
TR.ActiveTrainer.forward_train = fwd

tmp = pathlib.Path(tempfile.mkdtemp())
tree, args, fset = T._file_sets(tmp)
for tag, mk in (('file', lambda: fset), ('resident', lambda: T._resident_twin(args, tree)), ('resident2', lambda: T._resident_twin(args, tree))):
    print(tag, flush=True)
    r = T._round(args, mk(), tag)
    print('  losses', r['losses'], flush=True)
