"""captioning/modules/losses.StructureLosses against outputs of the REAL reference class
(tests/golden/structure_losses.npz, written by ``tests/golden/make_golden.py struct`` from
/root/reference/captioning/modules/losses.py:40-202): every log-probability structure loss type, with and without the entropy
reward, reductions 'mean' and 'none', loss AND gradient w.r.t. the dense log-probabilities.  CPU tensors: the criteria are
plain autograd over ``select_logp`` (the GPU tests run the same class on rollout outputs with the sparse gradient)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'imagecaptioning', 'pytorch_amd'))
Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'structure_losses.npz'))
CASES = sorted({k[:-5] for k in Z.files if k.endswith('_loss')})


@pytest.mark.parametrize('case', CASES)
def test_structure_loss_matches_the_reference(case):
    from captioning.modules import losses as L
    lt, e, red = case.rsplit('_', 2)
    ew = int(e[1:]) / 10.0
    B, n = int(Z['B']), int(Z['n'])
    opt = argparse.Namespace(structure_loss_type=lt, train_sample_n=n, entropy_reward_weight=ew, self_cider_reward_weight=0,
                             cider_reward_weight=1)
    scores = torch.from_numpy(Z['scores'])
    saved = L.get_scores
    L.get_scores = lambda data_gts, gen_result, o, as_tensor=False: scores.clone()
    try:
        raw = lt in L.StructureLosses.LOGIT_TYPES          # r4: the margin types read the logits themselves (output_logsoftmax = 0)
        x = (torch.from_numpy(Z['logits']).clone() if raw else torch.log_softmax(torch.from_numpy(Z['logits']), 2)).requires_grad_(True)
        o = L.StructureLosses(opt)(x, torch.from_numpy(Z['seq']), [None] * B, reduction=red)
    finally:
        L.get_scores = saved
    loss = o['loss']
    ref = torch.from_numpy(Z[case + '_loss'])
    # (equal_nan: an image whose n samples all score the same makes the rescaled costs of 'risk' / 'softmax_margin' 0/0 in the
    #  reference too -- rows 3..5 of the fixture)
    assert loss.shape == ref.shape and torch.allclose(loss, ref, rtol=1e-10, atol=1e-12, equal_nan=True), (loss, ref)
    w = torch.linspace(0.5, 1.5, loss.numel(), dtype=torch.float64).view_as(loss) if red == 'none' else None
    (loss if w is None else (loss * w).sum()).backward()
    assert torch.allclose(x.grad, torch.from_numpy(Z[case + '_grad']), rtol=1e-9, atol=1e-12, equal_nan=True)
    assert torch.allclose(o['reward'], torch.from_numpy(Z[case + '_reward']))


def test_cases_cover_every_structure_loss_type():
    from captioning.modules import losses as L
    assert {c.split('_e')[0] for c in CASES} == set(L.StructureLosses.LOGPROB_TYPES + L.StructureLosses.LOGIT_TYPES)
    crit = L.StructureLosses(argparse.Namespace(structure_loss_type='policy_gradient', train_sample_n=2))
    with pytest.raises(NotImplementedError):
        crit(torch.zeros(2, 3, 4), torch.ones(2, 3, dtype=torch.long), [None])


CRIT = np.load(os.path.join(ROOT, 'tests', 'golden', 'criteria.npz'))


@pytest.mark.parametrize('name', ['lm', 'ls', 'rl'])
@pytest.mark.parametrize('red', ['mean', 'none'])
def test_criteria_match_the_reference_classes(name, red):
    """RewardCriterion / LanguageModelCriterion / LabelSmoothing(0.2) (losses.py:18-37, 204-265) against the reference classes'
    own outputs (tests/golden/criteria.npz, ``make_golden.py crit``): both reductions (``'none'`` feeds drop_worst,
    train.py:187-191), [B, n, T] targets, targets and masks longer than the input, loss and dense gradient."""
    from captioning.modules import losses as L
    x = torch.log_softmax(torch.from_numpy(CRIT['logits']), 2).requires_grad_(True)
    tgt, mask = torch.from_numpy(CRIT['target']), torch.from_numpy(CRIT['mask'])
    N = x.shape[0]
    if name == 'lm':
        loss = L.LanguageModelCriterion()(x, tgt, mask, reduction=red)
    elif name == 'ls':
        loss = L.LabelSmoothing(smoothing=0.2)(x, tgt.view(N, -1), mask.view(N, -1), reduction=red)
    else:
        loss = L.RewardCriterion()(x, torch.from_numpy(CRIT['seq']), torch.from_numpy(CRIT['reward']), reduction=red)
    ref = torch.from_numpy(CRIT['%s_%s_loss' % (name, red)])
    assert loss.shape == ref.shape and torch.allclose(loss, ref, rtol=1e-10, atol=1e-12), (loss, ref)
    w = torch.linspace(0.5, 1.5, loss.numel(), dtype=torch.float64).view_as(loss) if red == 'none' else None
    (loss if w is None else (loss * w).sum()).backward()
    assert torch.allclose(x.grad, torch.from_numpy(CRIT['%s_%s_grad' % (name, red)]), rtol=1e-9, atol=1e-12)
