"""BvSB + region-level class balancing without the "undefined" channel (VOC form) -- reference
``active_selection/my_bvsb_clsbal_v2.py``: ``num_classes`` channels, no ban."""
from . import my_bvsb_clsbal_v2_banignore


class RegionSelector(my_bvsb_clsbal_v2_banignore.RegionSelector):
    extra_channels = 0
    ban_ignore = False
