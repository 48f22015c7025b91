"""Rollout-level parity of the HIP UpDown path on a real MI355X.

* against the COMMITTED GOLDEN FIXTURES produced by the real reference (tests/golden/updown_tiny.npz):
  greedy tokens exact, log-probs / losses / every parameter gradient within fp32 tolerance;
* against the oracle (oracle/att_lstm.py, itself pinned to those fixtures) at the BASELINE sizes
  (R=E=1000, A=512, V1=9488, K=36, L=20) with injected dropout masks and Gumbel noise.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def eng():
    from imagecaptioning.pytorch_amd import updown_engine
    return updown_engine


def load_golden():
    z = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')}
    return z, P


def to_dev(P):
    return {k: v.detach().to(DEV).contiguous() for k, v in P.items()}


def alloc_grads(Pd):
    return {k: torch.full_like(v, float('nan')) for k, v in Pd.items()}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_golden_greedy_token_exact(tag):
    E = eng()
    z, P = load_golden()
    Pd = to_dev(P)
    fc, att = torch.from_numpy(z['fc']).to(DEV), torch.from_numpy(z['att']).to(DEV)
    am = torch.from_numpy(z['att_masks']).to(DEV) if tag == 'mask' else None
    pr = E.prepare(Pd, fc, att, am)
    ro = E.Rollout(Pd, pr, n=1, T=8, mode='greedy')
    seq, slp = ro.run()
    assert np.array_equal(seq.cpu().numpy(), z['greedy_seq_' + tag])
    np.testing.assert_allclose(slp.cpu().numpy(), z['greedy_logp_' + tag], rtol=2e-5, atol=5e-6)


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_golden_teacher_forced_xe_and_grads(tag):
    from oracle import att_lstm as O
    E = eng()
    z, P = load_golden()
    Pd = to_dev(P)
    fc, att = torch.from_numpy(z['fc']).to(DEV), torch.from_numpy(z['att']).to(DEV)
    am = torch.from_numpy(z['att_masks']).to(DEV) if tag == 'mask' else None
    labels, masks = torch.from_numpy(z['labels']), torch.from_numpy(z['masks'])
    B, n, T1 = labels.shape
    inp = labels[..., :-1].reshape(B * n, -1).contiguous()
    Tfull = inp.shape[1]
    T_eff = Tfull
    for i in range(1, Tfull):                          # AttModel.py:158 early break, decided on the host ONCE
        if int(inp[:, i].sum()) == 0:
            T_eff = i
            break
    assert T_eff < Tfull
    pr = E.prepare(Pd, fc, att, am)
    ro = E.Rollout(Pd, pr, n=n, T=T_eff, L=Tfull, forced=inp.to(DEV), teacher=True)
    _, logp = ro.run()
    np.testing.assert_allclose(logp.cpu().numpy(), z['xe_logp_' + tag], rtol=2e-5, atol=5e-6)
    # loss + its gradient w.r.t. the dense log-probs by the criterion (host torch, tiny), then HIP BPTT
    lp = logp.detach().cpu().requires_grad_(True)
    loss = O.lm_criterion(lp, labels[..., 1:], masks[..., 1:])
    np.testing.assert_allclose(loss.item(), z['xe_loss_' + tag], rtol=1e-5)
    loss.backward()
    grads = alloc_grads(Pd)
    d_fc, d_att, d_p_att = ro.backward(lp.grad.to(DEV), grads)
    E.prepare_backward(Pd, pr, d_fc, d_att, d_p_att, grads)
    torch.cuda.synchronize()
    for k in Pd:
        ref = z['xe_grad_%s.%s' % (tag, k)]
        np.testing.assert_allclose(grads[k].cpu().numpy(), ref, rtol=5e-4, atol=1e-6 + 2e-5 * np.abs(ref).max(), err_msg=k)


def test_golden_sampled_forced_reward_grads():
    from oracle import att_lstm as O
    E = eng()
    z, P = load_golden()
    Pd = to_dev(P)
    fc, att, am = (torch.from_numpy(z[k]).to(DEV) for k in ('fc', 'att', 'att_masks'))
    forced = torch.from_numpy(z['sample_seq']).to(DEV)
    pr = E.prepare(Pd, fc, att, am)
    ro = E.Rollout(Pd, pr, n=2, T=8, mode='forced', forced=forced)
    seq, slp = ro.run()
    assert np.array_equal(seq.cpu().numpy(), z['sample_seq'])
    np.testing.assert_allclose(slp.cpu().numpy(), z['sample_logp'], rtol=2e-5, atol=5e-6)
    lp = slp.detach().cpu().requires_grad_(True)
    loss = O.reward_criterion(lp, seq.cpu(), torch.from_numpy(z['sample_reward']))
    np.testing.assert_allclose(loss.item(), z['rl_loss'], rtol=1e-5)
    loss.backward()
    grads = alloc_grads(Pd)
    d_fc, d_att, d_p_att = ro.backward(lp.grad.to(DEV), grads)
    E.prepare_backward(Pd, pr, d_fc, d_att, d_p_att, grads)
    for k in Pd:
        ref = z['rl_grad.' + k]
        np.testing.assert_allclose(grads[k].cpu().numpy(), ref, rtol=5e-4, atol=1e-6 + 2e-5 * np.abs(ref).max(), err_msg=k)


from shapes import full_size_params  # noqa: E402


def test_full_size_greedy_vs_oracle():
    """BASELINE shapes (36x2048 feats, R=1000, V1=9488, L=20): greedy decode of 10 images must be
    token-id-exact against the fp32 oracle; a near-tie flip is reported with its log-prob gap."""
    from oracle import att_lstm as O
    E = eng()
    P = full_size_params()
    g = torch.Generator().manual_seed(1)
    B, K, L = 10, 36, 20
    fc = (torch.randn(B, 2048, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, K, 2048, generator=g) * 0.5).clamp_min(0)
    with torch.no_grad():
        seq_ref, slp_ref = O.rollout(P, fc, att, None, method='greedy', max_len=L)
    Pd = to_dev(P)
    pr = E.prepare(Pd, fc.to(DEV), att.to(DEV), None)
    ro = E.Rollout(Pd, pr, n=1, T=L, mode='greedy')
    seq, slp = ro.run()
    seq_c = seq.cpu()
    if not torch.equal(seq_c, seq_ref):
        bad = (seq_c != seq_ref).nonzero()[0]
        r, t = int(bad[0]), int(bad[1])
        top2 = torch.topk(slp_ref[r, t], 2)[0]
        pytest.fail('first divergence row %d step %d: ref %d got %d, ref top-2 gap %.3e'
                    % (r, t, int(seq_ref[r, t]), int(seq_c[r, t]), float(top2[0] - top2[1])))
    assert float((slp.cpu() - slp_ref).abs().max()) < 1e-4


def test_full_size_scst_sample_and_grads_vs_oracle():
    """bs10 x sample_n 5, dropout 0.5 masks and Gumbel noise injected on both sides: sampled tokens
    exact, selected log-probs and RewardCriterion loss within 1e-4, gradients within 1e-3 relative."""
    from oracle import att_lstm as O
    E = eng()
    P = full_size_params(seed=99)
    g = torch.Generator().manual_seed(2)
    B, n, K, L = 4, 5, 36, 8          # oracle backward at N=50,L=20 takes minutes on CPU; same code path
    N = B * n
    R = Ed = 1000
    V1 = 9488
    fc = (torch.randn(B, 2048, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, K, 2048, generator=g) * 0.5).clamp_min(0)
    drops = O.make_drops(0.5, B, K, N, L, Ed, R, g)
    gum = -torch.log(-torch.log(torch.rand(L, N, V1, generator=g).clamp_min(1e-20)))
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    seq_ref, slp_ref = O.rollout(Pg, fc, att, None, method='sample', sample_n=n, temperature=1.0, max_len=L, drops=drops,
                                 gumbel=gum)
    reward = torch.randn(N, 1, generator=g).repeat(1, L)
    loss_ref = O.reward_criterion(slp_ref, seq_ref, reward)
    loss_ref.backward()
    Pd = to_dev(P)
    d = lambda t: t.to(DEV).contiguous()                                 # noqa: E731
    pr = E.prepare(Pd, d(fc), d(att), None, drop_fc=d(drops.fc), drop_att=d(drops.att))
    ro = E.Rollout(Pd, pr, n=n, T=L, mode='sample', temperature=1.0, drop_xt=d(drops.xt), drop_out=d(drops.out),
                   gumbel=d(gum))
    seq, slp = ro.run()
    assert torch.equal(seq.cpu(), seq_ref), 'sampled tokens differ'
    assert float((slp.cpu() - slp_ref.detach()).abs().max()) < 1e-4
    lp = slp.detach().cpu().requires_grad_(True)
    loss = O.reward_criterion(lp, seq.cpu(), reward)
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    loss.backward()
    grads = alloc_grads(Pd)
    d_fc, d_att, d_p_att = ro.backward(d(lp.grad), grads)
    E.prepare_backward(Pd, pr, d_fc, d_att, d_p_att, grads)
    for k in Pd:
        if k == 'core.attention.alpha_net.bias':
            continue       # mathematically zero (softmax shift invariance): pure rounding noise on both sides
        assert rel(grads[k], Pg[k].grad) < 1e-3, k


def test_rollout_edge_cases_immediate_eos_and_mixed_lengths():
    """Edge cases of AttModel._sample (:340-350): rows that emit EOS at step 0 (only step 0 carries log-probs, everything
    after is pad / zero rows), rows that never finish, and a mixture, against the oracle on the golden weights -- plus
    the BPTT through such a batch (finished steps contribute no gradient)."""
    from oracle import att_lstm as O
    E = eng()
    z, P = load_golden()
    Pd = to_dev(P)
    fc, att = (torch.from_numpy(z[k]) for k in ('fc', 'att'))
    B, n, L = fc.shape[0], 2, 8
    N = B * n
    g = torch.Generator().manual_seed(4)
    forced = torch.randint(1, 30, (N, L), generator=g)
    forced[0] = 0                      # EOS immediately
    forced[1, 3:] = 0                  # stops after 3 words
    forced[2, 0] = 5
    forced[2, 1:] = 0                  # one word
    # forced[3:] never emit EOS
    seq_o, logp_o = O.rollout(P, fc, att, None, sample_n=n, max_len=L, forced=forced)
    pr = E.prepare(Pd, fc.to(DEV), att.to(DEV), None)
    ro = E.Rollout(Pd, pr, n=n, T=L, mode='forced', forced=forced.to(DEV))
    seq, slp = ro.run()
    assert np.array_equal(seq.cpu().numpy(), seq_o.numpy())
    assert seq[0].sum() == 0 and float(slp[0, 1:].abs().max()) == 0.0 and float(slp[0, 0].abs().max()) > 0
    assert float(slp[2, 2:].abs().max()) == 0.0
    np.testing.assert_allclose(slp.cpu().numpy(), logp_o.detach().numpy(), rtol=2e-5, atol=5e-6)
    # gradient of a reward criterion through the ragged batch
    reward = torch.randn(N, generator=g)[:, None].expand(N, L).contiguous()
    lp = slp.detach().cpu().requires_grad_(True)
    O.reward_criterion(lp, seq.cpu(), reward).backward()
    grads = alloc_grads(Pd)
    d_fc, d_att, d_p_att = ro.backward(lp.grad.to(DEV), grads)
    E.prepare_backward(Pd, pr, d_fc, d_att, d_p_att, grads)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    _, logp_g = O.rollout(Pg, fc, att, None, sample_n=n, max_len=L, forced=forced)
    O.reward_criterion(logp_g, seq_o, reward).backward()
    for k in Pd:
        ref = Pg[k].grad.numpy()
        np.testing.assert_allclose(grads[k].cpu().numpy(), ref, rtol=1e-3, atol=1e-6 + 5e-5 * np.abs(ref).max(), err_msg=k)


@pytest.mark.parametrize('smoothing', [0.0, 0.2])
def test_full_size_xe_loss_and_grads_vs_oracle(smoothing):
    """BASELINE configs[1] shapes (R=E=1000, V1=9488, K=36), teacher forcing with dropout 0.5 masks injected on both
    sides, LanguageModelCriterion and LabelSmoothing (losses.py:204-265): loss within 1e-4, gradients within 1e-3
    relative.  Rows of different lengths: the padded tail carries no loss and no gradient."""
    from oracle import att_lstm as O
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion, LabelSmoothing
    E = eng()
    P = full_size_params(seed=7)
    g = torch.Generator().manual_seed(3)
    B, n, K, L = 3, 2, 36, 7
    N, R, Ed, V1 = B * n, 1000, 1000, 9488
    fc = (torch.randn(B, 2048, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, K, 2048, generator=g) * 0.5).clamp_min(0)
    T = L + 1                                             # inputs [BOS, w_0 .. w_{L-1}]
    labels = torch.zeros(N, L + 2, dtype=torch.long)
    masks = torch.zeros(N, L + 2)
    for r in range(N):
        ln = L if r == 0 else int(torch.randint(2, L + 1, (1,), generator=g))     # row 0 full length: no early break
        labels[r, 1:ln + 1] = torch.randint(1, V1, (ln,), generator=g)
        masks[r, :ln + 2] = 1
    drops = O.make_drops(0.5, B, K, N, T, Ed, R, g)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    logp_ref = O.forward_teacher(Pg, fc, att, labels[:, :-1].view(B, n, -1), None, drops)
    crit_ref = (lambda lp: O.lm_criterion(lp, labels[:, 1:], masks[:, 1:])) if smoothing == 0 else \
        (lambda lp: O.label_smoothing_criterion(lp, labels[:, 1:], masks[:, 1:], smoothing))
    loss_ref = crit_ref(logp_ref)
    loss_ref.backward()
    Pd = to_dev(P)
    d = lambda t: t.to(DEV).contiguous()                                 # noqa: E731
    pr = E.prepare(Pd, d(fc), d(att), None, drop_fc=d(drops.fc), drop_att=d(drops.att))
    ro = E.Rollout(Pd, pr, n=n, T=T, L=T, mode='forced', forced=d(labels[:, :-1]), teacher=True, drop_xt=d(drops.xt),
                   drop_out=d(drops.out))
    _, slp = ro.run()
    assert float((slp.cpu() - logp_ref.detach()).abs().max()) < 1e-4
    lp = slp.detach().requires_grad_(True)
    crit = LanguageModelCriterion() if smoothing == 0 else LabelSmoothing(smoothing=smoothing)
    loss = crit(lp, d(labels[:, 1:]), d(masks[:, 1:]))
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
    loss.backward()
    grads = alloc_grads(Pd)
    d_fc, d_att, d_p_att = ro.backward(lp.grad, grads)
    E.prepare_backward(Pd, pr, d_fc, d_att, d_p_att, grads)
    for k in Pd:
        if k == 'core.attention.alpha_net.bias':
            continue
        assert rel(grads[k], Pg[k].grad) < 1e-3, k
