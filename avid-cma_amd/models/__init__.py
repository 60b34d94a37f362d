"""MI355X-native drop-in for the reference's ``models`` package (models/__init__.py:8-10).

``utils/main_utils.py:77`` resolves ``models.__dict__[cfg['arch']]`` — the registry names
(``av_wrapper``, ``R2Plus1D``, ``Conv2D``) and constructor signatures are kept.
"""
from .video import *  # noqa: F401,F403
from .audio import *  # noqa: F401,F403
from .av_wrapper import *  # noqa: F401,F403
