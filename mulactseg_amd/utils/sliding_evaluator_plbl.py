"""Sliding-window inference returning (features [256,H,W], scores [C,H,W]) -- reference
``utils/sliding_evaluator_plbl.py:8-157``; see ``sliding_evaluator.py`` for what differs from the reference."""
from . import sliding_evaluator


class SlidingEval(sliding_evaluator.SlidingEval):
    with_features = True
