"""PixBal + ban of regions dominated by the "undefined" class -- reference
``active_selection/my_bvsb_predclsbal_pwr_banignore.py`` (Cityscapes form: the model predicts
``num_classes + 1`` channels; regions whose arg-max-class histogram peaks at the last channel get
score 0)."""
from . import my_bvsb_predclsbal_pwr


class RegionSelector(my_bvsb_predclsbal_pwr.RegionSelector):
    extra_channels = 1        # (my_bvsb_predclsbal_pwr_banignore.py:32,68)
    ban_ignore = True
