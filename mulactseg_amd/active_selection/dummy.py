"""No-op selector kept for driver compatibility -- reference ``active_selection/dummy.py``."""


class RegionSelector(object):
    def __init__(self, args):
        pass

    def select_next_batch(self, trainer, active_set, selection_count):
        print("dummy selection: pass")
