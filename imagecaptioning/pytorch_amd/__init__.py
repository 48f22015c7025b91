"""MI355X-native (gfx950 / CDNA4) backend for the caption-decoding hot path of
ruotianluo/ImageCaptioning.pytorch: hand-written HIP kernels behind the C ABI of include/capmi.h,
driven from a host-side mirror of the reference's ``captioning`` package (``captioning/`` here).

Every compute submodule imports ``_lib``, which loads ``libcapmi.so`` and raises if it is missing or
stale: there is no CPU fallback.  (The package ``__init__`` itself stays import-light so that
``python -m imagecaptioning.pytorch_amd.build`` can run before the library exists.)
"""


def load_library():
    """Load (and return) the ctypes handle of libcapmi.so; raises ImportError if it is not built."""
    from . import _lib
    return _lib.lib
