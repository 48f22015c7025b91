"""Host-side mirror of the reference's ``captioning`` package for the caption-decoding hot path,
backed by libcapmi (MI355X HIP kernels).  Same module paths, class names, call signatures and
``state_dict`` keys as /root/reference/captioning, so ``tools/train.py`` / ``tools/eval.py``-style
callers can switch backends by putting this directory's parent on ``sys.path``.

The package works under both names -- ``imagecaptioning.pytorch_amd.captioning`` and, when
``imagecaptioning/pytorch_amd`` is on ``sys.path``, plain ``captioning`` (shadowing the reference's) -- so it
reaches the backend through the absolute name ``imagecaptioning.pytorch_amd`` and makes the repository root
importable if it is not yet.
"""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _ROOT not in _sys.path:
    _sys.path.append(_ROOT)
