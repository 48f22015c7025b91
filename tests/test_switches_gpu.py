"""The library's per-process switches (read once by libcapmi, hence child processes): every alternative path they select must pass
the same parity tests as the default path.

* CAPMI_BWD_SIDE=2 -- the BPTT's weight-gradient GEMMs in two time chunks, the later one on a side stream beside the time loop
  (csrc/rollout.hip; off by default, profiles/r05_scst_overlap.md): gradients of the UpDown fixtures and of the full-size SCST case.
* CAPMI_BATCHED_XT=0 -- the teacher-forced token-embedding projection inside every per-step gate GEMM (the pre-r5 path) instead of
  one GEMM over all steps: the reference's XE fixtures, tiny and at bs64 x 5.
* CAPMI_X3_TILE=256 -- every bf16x3 fat GEMM on 256 x 128 tiles, also where the planner would not choose them.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _child(env, args):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'] + args, cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and ' failed' not in r.stdout, r.stdout[-2000:]


def test_side_stream_weight_gradients_pass_the_gradient_fixtures():
    _child({'CAPMI_BWD_SIDE': '2'}, ['tests/test_updown_gpu.py', 'tests/test_full_size_parity_gpu.py::test_updown_c3_shape_scst_tokens_loss_and_gradients_vs_fixture',
                                     'tests/test_full_size_parity_gpu.py::test_updown_xe_bs10x5_vs_the_reference_itself'])


def test_per_step_token_embedding_projection_passes_the_xe_fixtures():
    _child({'CAPMI_BATCHED_XT': '0'}, ['tests/test_full_size_parity_gpu.py::test_updown_xe_bs10x5_vs_the_reference_itself',
                                       'tests/test_full_size_parity_gpu.py::test_updown_xe_at_its_own_batch_bs64x5_vs_the_reference_itself',
                                       'tests/test_model_api_gpu.py'])


def test_wide_tiles_everywhere_pass_the_big_batch_fixtures():
    _child({'CAPMI_X3_TILE': '256'}, ['tests/test_full_size_parity_gpu.py::test_updown_xe_at_its_own_batch_bs64x5_vs_the_reference_itself',
                                      'tests/test_full_size_parity_gpu.py::test_transformer_and_aoa_at_the_baseline_batch_vs_the_reference_itself'])
