"""GPU-side pieces of the reference's ``datasets`` package.

Only the audio front end lives here (``datasets.gpu_audio.LogSpectrogram``, the batched HIP replacement of
``datasets/preprocessing.py:158-186``).  Like ``utils``, this package extends its ``__path__`` over every other
``datasets`` directory on ``sys.path``, so with this directory listed before the reference checkout the
reference's own ``datasets.video_db`` / ``datasets.preprocessing`` ... keep resolving to its files.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
