"""Selector that selects nothing: lets ``train_AL.py`` run a round on the current labelled set
(plugin surface of the reference's ``active_selection/dummy.py``: ``RegionSelector(args).select_next_batch``)."""
import logging

_LOG = logging.getLogger(__name__)


class RegionSelector:
    """Same constructor and call signature as every other selector; the active set is left untouched."""

    def __init__(self, args):
        self.args = args
        self.rounds_skipped = 0

    def select_next_batch(self, trainer, active_set, selection_count):
        self.rounds_skipped += 1
        _LOG.info("no-op selector: %d regions requested, none added (call %d)", selection_count, self.rounds_skipped)
        return None
