"""Start / end-of-experiment reporting -- the reference's ``utils/mylog.py:8-46``: ``init_logging`` prepares the per-round IoU table
the trainers fill (``trainer/base.py:eval``), ``finalization`` writes the experiment report into the log.  wandb is optional here
(``args.wandb`` is used when the driver sets it)."""
import logging
from datetime import datetime


def timediff(t_start, t_end):
    """``'{h}h {m}m {s}s'`` of the elapsed time (hours wrap at a day, as ``relativedelta(...).hours`` does)."""
    s = int((t_end - t_start).total_seconds())
    return '{h}h {m}m {s}s'.format(h=(s // 3600) % 24, m=(s // 60) % 60, s=s % 60)


def init_logging(args):
    """``args.wandb_iou_table``: one row, a column ``round_v_miou`` and one column per round 0..max_iterations (:25-32)."""
    import pandas as pd
    cols = {"round_v_miou": [""]}
    for i in range(args.max_iterations + 1):
        cols["round-{}".format(i)] = [""]
    args.wandb_iou_table = pd.DataFrame(data=cols)


def log_final(t_start, val_result, logger, args):
    t_end = datetime.now()
    logger.info("%s Experiment Report %s" % ('%' * 20, '%' * 20))
    logging.info("0. AL Methods: %s" % args.active_method)
    logging.info("1. Takes: %s" % timediff(t_start, t_end))
    logging.info("2. Log dir: %s (with selection json & model checkpoint)" % args.model_save_dir)
    logging.info("3. Validation mIoU (Be sure to submit to google form)")
    for selection_iter in range(args.init_iteration, args.max_iterations + 1):
        logging.info("AL %d: %s" % (selection_iter, val_result[selection_iter]))
    logger.info("%s Experiment End %s" % ('%' * 20, '%' * 20))


def finalization(t_start, val_result, logger, args):
    log_final(t_start, val_result, logger, args)
