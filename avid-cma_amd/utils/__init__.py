"""Drop-in for the hot-path members of the reference's ``utils`` package.

Only ``utils.alias_method`` and ``utils.distributed_utils`` are on the hot path.
The reference's ``utils/__init__.py`` is empty, so this package extends its ``__path__`` over every
other ``utils`` directory on ``sys.path``: with this directory listed BEFORE the reference checkout,
``utils.alias_method`` resolves here while ``utils.main_utils`` / ``utils.logger`` still resolve to
the reference's own (unmodified) files — which is what lets ``main-avid.py`` run unchanged.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
