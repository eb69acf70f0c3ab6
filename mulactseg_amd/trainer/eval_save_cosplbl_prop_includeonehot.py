"""The production stage-2 generator (``script/open_source/train_city_mul_res50.sh:49``): as
``eval_save_cosplbl_prop`` but every selected superpixel takes part, one-hot ones included --
reference ``trainer/eval_save_cosplbl_prop_includeonehot.py`` (a 6-line delta, :167-172)."""
from . import eval_save_cosplbl_prop


class ActiveTrainer(eval_save_cosplbl_prop.ActiveTrainer):
    include_onehot = True
