"""Early exit of free-running rollouts (AttModel.py:349-350 `if unfinished.sum() == 0: break`): with a model that
emits its EOS within a few steps the driver stops enqueuing, and tokens, log-probs and gradients are those of the run that
enqueues all L steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_early_exit_equals_full_length_rollout():
    from imagecaptioning.pytorch_amd import updown_engine as E
    from shapes import full_size_params
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    B, n, K, L = 10, 5, 36, 20
    P = {k: v.to(dev).contiguous() for k, v in full_size_params(seed=5).items()}
    P['logit.bias'] = P['logit.bias'].clone()
    P['logit.bias'][0] += 12.0                     # EOS (token 0) takes ~94 % of the mass per step: every row ends within a few steps
    fc = torch.randn(B, 2048, device=dev).clamp_min(0)
    att = torch.randn(B, K, 2048, device=dev).clamp_min(0)
    pr = E.prepare(P, fc, att, None)
    N, V1, R, Em = B * n, P['logit.weight'].shape[0], 1000, 1000
    gum = torch.rand(L, N, V1, device=dev).clamp_min(1e-12).log().neg().log().neg()
    drop_xt = (torch.rand(L, N, Em, device=dev) < 0.5).float() * 2
    drop_out = (torch.rand(L, N, R, device=dev) < 0.5).float() * 2
    out = {}
    for ee in (0, 2, 4):
        ro = E.Rollout(P, pr, n=n, T=L, mode='sample', gumbel=gum, drop_xt=drop_xt, drop_out=drop_out, early_exit=ee,
                       early_exit_from=0 if ee == 2 else 4)
        seq, slp = ro.run()
        torch.cuda.synchronize()
        grads = {k: torch.zeros_like(P[k]) for k in E.PARAM_KEYS}
        gsl = torch.zeros_like(slp)
        mask = torch.cat([torch.ones(N, 1, device=dev, dtype=torch.bool), seq[:, :-1] > 0], 1)      # RewardCriterion's mask
        gsl.scatter_(2, seq.unsqueeze(-1), (-0.3 * mask.float()).unsqueeze(-1))
        d = ro.backward(gsl, grads)
        torch.cuda.synchronize()
        out[ee] = (seq.clone(), slp.clone(), ro.sel_logp.clone(), {k: v.clone() for k, v in grads.items()}, [t.clone() for t in d],
                   ro.steps_run)
    full = out[0]
    last = int((full[0] > 0).any(0).nonzero().max()) + 1 if bool((full[0] > 0).any()) else 0      # steps with a live token
    assert full[5] == L and last <= 6, (full[5], last)
    for ee in (2, 4):
        o = out[ee]
        assert o[5] < L, 'the rollout did not stop early'
        assert o[5] >= last + 1                  # the step at which the last row emits EOS is always run
        assert torch.equal(o[0], full[0]) and torch.equal(o[1], full[1]) and torch.equal(o[2], full[2])
        for k in full[3]:
            if k == 'core.attention.alpha_net.bias':      # shift-invariant softmax: rounding noise around an exact 0
                continue
            ref = full[3][k]
            assert float((o[3][k] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-12, k
        for x, y in zip(o[4], full[4]):
            assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max()) + 1e-12
    assert out[2][5] <= 8 + (last > 4) * 2, out[2][5]      # EOS by step <= 6, checks every 2 steps: <= 8-10 steps of kernels


def test_early_exit_never_fires_on_a_model_that_never_ends():
    from imagecaptioning.pytorch_amd import updown_engine as E
    from shapes import full_size_params
    dev = torch.device('cuda:0')
    P = {k: v.to(dev).contiguous() for k, v in full_size_params(seed=6).items()}
    P['logit.bias'] = P['logit.bias'].clone()
    P['logit.bias'][0] -= 50.0                     # EOS never drawn
    fc = torch.randn(4, 2048, device=dev).clamp_min(0)
    att = torch.randn(4, 36, 2048, device=dev).clamp_min(0)
    pr = E.prepare(P, fc, att, None)
    ro = E.Rollout(P, pr, n=5, T=20, mode='sample', seed=3, early_exit=4)
    seq, _ = ro.run()
    torch.cuda.synchronize()
    assert ro.steps_run == 20 and bool((seq > 0).all())


def test_a_rollout_object_can_be_run_again_after_an_early_exit():
    """run() used to leave r.T at the number of steps the early exit had enqueued, so a SECOND run() on the same object
    was capped at that length.  With weights that end captions early on the first run and never on the second, the second run must
    enqueue all T steps again."""
    from imagecaptioning.pytorch_amd import updown_engine as E
    from shapes import full_size_params
    dev = torch.device('cuda:0')
    P = {k: v.to(dev).contiguous() for k, v in full_size_params(seed=6).items()}
    P['logit.bias'] = P['logit.bias'].clone()
    P['logit.bias'][0] += 12.0
    fc = torch.randn(4, 2048, device=dev).clamp_min(0)
    att = torch.randn(4, 36, 2048, device=dev).clamp_min(0)
    pr = E.prepare(P, fc, att, None)
    ro = E.Rollout(P, pr, n=5, T=20, mode='sample', seed=3, early_exit=2, early_exit_from=0)
    ro.run()
    torch.cuda.synchronize()
    assert ro.steps_run < 20
    P['logit.bias'][0] -= 62.0                     # the same weight tensors, now a model that never ends (the structs hold pointers)
    seq, _ = ro.run()
    torch.cuda.synchronize()
    assert ro.steps_run == 20 and bool((seq > 0).all())
