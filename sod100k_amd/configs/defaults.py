"""Counterpart of the reference's ``configs/defaults.py`` (``from .defaults import _C as cfg``, configs/__init__.py:2):
the default node lives in ``sod100k_amd/configs/__init__.py`` (yacs-free ``CfgNode``); this module re-exports it under
the reference's names."""
from . import cfg as _C, defaults as get_cfg_defaults  # noqa: F401
