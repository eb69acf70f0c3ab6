"""Trainer plugins with the reference's surface: module ``trainer.<method>`` exposing
``class ActiveTrainer(args, logger, selection_iter)`` with ``.net``, ``.device``, ``.model_save_dir``,
``.selection_iter``, ``.load_checkpoint``, ``.train``, ``.eval`` (reference ``trainer/base.py``,
``trainer/active.py`` and the production chain listed in SURVEY.md section 2.1 #4)."""
