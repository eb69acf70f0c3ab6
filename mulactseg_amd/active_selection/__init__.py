"""Region-selector plugins with the reference's surface: module ``active_selection.<name>`` exposing
``class RegionSelector(args)`` with ``calculate_scores(trainer, pool_set)`` and
``select_next_batch(trainer, active_set, selection_count)`` (reference ``active_selection/base.py``)."""
