"""Host-side mirror of the reference's ``captioning`` package for the caption-decoding hot path,
backed by libcapmi (MI355X HIP kernels).  Same module paths, class names, call signatures and
``state_dict`` keys as /root/reference/captioning, so ``tools/train.py`` / ``tools/eval.py``-style
callers can switch backends by putting this directory's parent on ``sys.path``.
"""
