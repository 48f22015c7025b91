"""CPU oracle for the caption-decoding hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``imagecaptioning/pytorch_amd``) never imports this package and fails loudly
when its HIP library is missing.

Contents
--------
``att_lstm.py``   plain PyTorch fp32 (CPU) restatement of the reference's
                  UpDown / NewFC decode step, teacher-forced forward, the
                  greedy / sampling rollout and the criteria
                  (``/root/reference/captioning/models/AttModel.py``,
                  ``FCModel.py``, ``CaptionModel.py``, ``modules/losses.py``).
                  PINNED: checked against outputs of the imported reference
                  itself (``tests/golden/*.npz`` made by
                  ``tests/golden/make_golden.py``).
``ciderd.py``     float64 restatement of CIDEr-D as published in
                  ``ruotianluo/cider`` (``pyciderevalcap/ciderD``) -- that
                  submodule is EMPTY under ``/root/reference`` and its pinned
                  commit is unknown, so this part is **parity unpinned**:
                  it is anchored only on the reference's call sites
                  (``captioning/utils/rewards.py:33-81``) and on hand-derived
                  known-answer tests.
``ciderd_c/``     the same arithmetic in plain C (gcc), used to cross-check
                  the Python restatement and as a faster CPU baseline.
"""
