"""MI355X-native drop-in for the reference's ``criterions`` package (criterions/__init__.py:7-8).

``utils/main_utils.py:233`` resolves ``criterions.__dict__[cfg['name']]`` -> ``AVID`` / ``AVID_CMA``.
"""
from .avid import *  # noqa: F401,F403
from .avid_cma import *  # noqa: F401,F403
