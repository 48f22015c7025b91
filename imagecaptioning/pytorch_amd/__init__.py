"""MI355X-native (gfx950 / CDNA4) backend for the caption-decoding hot path of
ruotianluo/ImageCaptioning.pytorch: hand-written HIP kernels behind the C ABI of include/capmi.h,
driven from a host-side mirror of the reference's ``captioning`` package (``captioning/`` here).

Importing this package loads ``libcapmi.so`` and raises if it is missing: there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fail loudly if the HIP library is absent)

__all__ = ['_lib']
