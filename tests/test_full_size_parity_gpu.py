"""BASELINE-size parity of the HIP paths (VERDICT r1 "weak" #2-#4): every model family at its configs/*.yml size against
the oracle or against fixtures computed by the reference itself -- the tiny golden fixtures cannot see size-dependent bugs
(LDS limits, split-K plans, tile edges).

* UpDown configs[2] shape (bs10 x 5, L=20, dropout 0.5): tokens, selected log-probs, RewardCriterion loss and every
  parameter gradient against ``tests/golden/updown_full_grads.npz`` (c3_*: oracle with injected masks / noise).
* UpDown teacher-forced XE at bs10 x 5, T=21 against the REAL reference's outputs (c2_* of the same file).
* Transformer (d=512, N=6, h=8, d_ff=2048) and AoA (R=1024, h=8, 6 refiner layers) against oracle/transformer.py and
  oracle/aoa.py run live on the same weights: log-probs <= 1e-4, gradients <= 1e-3 relative, greedy token-exact.
* NewFC at R=E=512, V1=9488 (configs/fc.yml) against oracle/att_lstm.py.
* StructureLosses 'new_self_critical' against the reference's ``nsc_loss`` fixture.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import shapes

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def to_dev(P):
    return {k: v.detach().to(DEV).contiguous() for k, v in P.items()}


def check_grads_against_fixture(z, prefix, grads, skip=()):
    for k, g in grads.items():
        if k in skip:
            continue
        want_norm = float(z['%s_gnorm.%s' % (prefix, k)])
        got_norm = float(g.double().norm())
        assert abs(got_norm - want_norm) <= 1e-3 * want_norm + 1e-9, (k, got_norm, want_norm)
        want = z['%s_gprobe.%s' % (prefix, k)]
        got = shapes.grad_probe(g).cpu().numpy()
        scale = float(g.abs().max())
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-3 * scale + 1e-9, err_msg=k)


def test_updown_c3_shape_scst_tokens_loss_and_gradients_vs_fixture():
    from imagecaptioning.pytorch_amd import updown_engine as E
    from oracle import att_lstm as O
    z = np.load(os.path.join(GOLDEN, 'updown_full_grads.npz'))
    P = shapes.full_size_params(seed=99)
    c = shapes.c3_case(seed=2)
    Pd = to_dev(P)
    d = lambda t: t.to(DEV).contiguous()                                 # noqa: E731
    dr = c['drops']
    pr = E.prepare(Pd, d(c['fc']), d(c['att']), None, drop_fc=d(dr.fc), drop_att=d(dr.att))
    ro = E.Rollout(Pd, pr, n=c['n'], T=c['L'], mode='sample', temperature=1.0, drop_xt=d(dr.xt), drop_out=d(dr.out),
                   gumbel=d(c['gumbel']))
    seq, slp = ro.run()
    assert np.array_equal(seq.cpu().numpy(), z['c3_seq']), 'sampled tokens differ from the oracle fixture'
    sel = slp.gather(2, seq.unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(sel, z['c3_sel_logp'], rtol=0, atol=1e-4)
    lp = slp.detach().cpu().requires_grad_(True)
    loss = O.reward_criterion(lp, seq.cpu(), c['reward'])
    assert abs(loss.item() - float(z['c3_loss'])) < 1e-4
    loss.backward()
    grads = {k: torch.full_like(v, float('nan')) for k, v in Pd.items()}
    d_fc, d_att, d_p_att = ro.backward(d(lp.grad), grads)
    E.prepare_backward(Pd, pr, d_fc, d_att, d_p_att, grads)
    # alpha_net.bias: mathematically zero (softmax shift invariance), pure rounding noise on both sides
    check_grads_against_fixture(z, 'c3', grads, skip=('core.attention.alpha_net.bias',))


def test_updown_xe_bs10x5_vs_the_reference_itself():
    """model API (captioning.models.setup + LanguageModelCriterion) on the weights / inputs the reference was run on"""
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion
    z = np.load(os.path.join(GOLDEN, 'updown_full_grads.npz'))
    model = models.setup(synthetic.updown_opt(drop_prob_lm=0.0))
    model.load_state_dict(shapes.full_size_params(seed=7))
    model = model.to(DEV)
    model.train()
    fc, att = shapes.feats(10, seed=3)
    labels, masks = shapes.c2_labels()
    labels, masks = labels.to(DEV), masks.to(DEV)
    logp = model(fc.to(DEV), att.to(DEV), labels[..., :-1], None)
    tgt = labels[..., 1:].reshape(-1, labels.shape[-1] - 1)
    got = logp.detach().gather(2, tgt[:, :logp.shape[1]].unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(got, z['c2_tgt_logp'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logp.detach()[0, :3].cpu().numpy(), z['c2_logp_row0'], rtol=0, atol=1e-4)
    loss = LanguageModelCriterion()(logp, labels[..., 1:], masks[..., 1:])
    assert abs(loss.item() - float(z['c2_loss'])) < 1e-4
    loss.backward()
    check_grads_against_fixture(z, 'c2', {k: p.grad for k, p in model.named_parameters()},
                                skip=('core.attention.alpha_net.bias',))


def _labels(B, n, L, V1, seed):
    g = torch.Generator().manual_seed(seed)
    labels = torch.zeros(B, n, L + 2, dtype=torch.long)
    masks = torch.zeros(B, n, L + 2)
    for b in range(B):
        for j in range(n):
            ln = L if (b == 0 and j == 0) else int(torch.randint(2, L + 1, (1,), generator=g))
            labels[b, j, 1:ln + 1] = torch.randint(1, V1, (ln,), generator=g)
            masks[b, j, :ln + 2] = 1
    return labels, masks


def _compare_model_with_oracle(model, forward_ref, greedy_ref, att, am, labels, masks, fc=None, relu_ties=None):
    """teacher-forced log-probs / XE loss / all gradients and the greedy decode of `model` (HIP) vs oracle callables.
    relu_ties: dict filled by the oracle's forward ({ffn prefix: bool [d_ff]}, oracle/transformer.py RELU_TIES): hidden units whose
    ReLU input is within 1e-4 of zero for some token.  With 2048 units x 24 tokens x 12 layers a handful always are, some within
    1e-6, and which side an fp32 implementation lands on is rounding: that unit's row of dW1 / db1 then differs by one token's
    whole term (seen as 1.6e-3 of the largest entry).  Those rows are held to 5e-2, every other row to 1e-3."""
    from oracle import att_lstm as O
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith('.pe'):
            v.requires_grad_(True)
    want = forward_ref(P)
    loss_w = O.lm_criterion(want, labels[..., 1:], masks[..., 1:])
    loss_w.backward()
    dv = lambda t: None if t is None else t.to(DEV)                    # noqa: E731
    logp = model(dv(fc), dv(att), labels[..., :-1].to(DEV), dv(am))
    assert float((logp.detach().cpu() - want.detach()).abs().max()) < 1e-4
    loss = LanguageModelCriterion()(logp, labels[..., 1:].to(DEV), masks[..., 1:].to(DEV))
    assert abs(loss.item() - loss_w.item()) < 1e-4
    model.zero_grad()
    loss.backward()
    worst = {}
    for k, p in model.named_parameters():
        ref = P[k].grad
        if float(ref.abs().max()) < 1e-12 or k.endswith('linears.1.bias'):
            # attention KEY bias: adds the same q.b to every score of a query, which the softmax cancels -- its gradient is
            # mathematically zero and pure rounding noise on both sides
            assert float(p.grad.abs().max()) < 1e-5 and float(ref.abs().max()) < 1e-5, k
            continue
        tie = None
        if relu_ties and (k.endswith('.w_1.weight') or k.endswith('.w_1.bias')):
            tie = relu_ties.get(k.rsplit('.w_1.', 1)[0])
        if tie is not None and bool(tie.any()):
            scale = float(ref.abs().max())
            err = (p.grad.detach().cpu().double() - ref.double()).abs()
            err = err.reshape(err.shape[0], -1).max(1)[0] / scale          # per hidden unit
            assert float(err[tie].max()) < 5e-2, (k, 'rows with a ReLU tie', float(err[tie].max()))
            assert int(tie.sum()) < 0.05 * tie.numel(), (k, int(tie.sum()))
            worst[k] = float(err[~tie].max())
        else:
            worst[k] = rel(p.grad, ref)
    bad = {k: v for k, v in worst.items() if v >= 1e-3}
    assert not bad, bad
    model.eval()
    with torch.no_grad():
        seq_w, slp_w = greedy_ref(P)
        seq, slp = model(dv(fc), dv(att), dv(am), opt={'sample_method': 'greedy', 'beam_size': 1}, mode='sample')
    seq_c = seq.cpu()
    if not torch.equal(seq_c, seq_w):
        bad = (seq_c != seq_w).nonzero()[0]
        r, t = int(bad[0]), int(bad[1])
        top2 = torch.topk(slp_w[r, t], 2)[0]
        pytest.fail('greedy diverges at row %d step %d (ref %d got %d), reference top-2 gap %.3e'
                    % (r, t, int(seq_w[r, t]), int(seq_c[r, t]), float(top2[0] - top2[1])))
    assert float((slp.cpu() - slp_w).abs().max()) < 1e-4


@pytest.mark.parametrize('masked', [False, True])
def test_transformer_baseline_size_vs_oracle(masked):
    """configs/transformer/transformer.yml sizes (d=512, d_ff=2048, h=8, N=6), TransformerModel.py:340-362"""
    from oracle import transformer as T
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    opt = synthetic.updown_opt(caption_model='transformer', input_encoding_size=512, rnn_size=2048, d_model=512, d_ff=2048,
                               N_enc=6, N_dec=6, num_att_heads=8, dropout=0.0, drop_prob_lm=0.0, seq_length=6, max_length=6)
    torch.manual_seed(11)
    model = models.setup(opt).to(DEV)
    model.train()
    B, n, L, K = 2, 2, 6, 36
    _, att = shapes.feats(B, K=K, seed=4)
    am = None
    if masked:
        am = torch.ones(B, K)
        am[0, 30:] = 0
        am[1, 17:] = 0
    labels, masks = _labels(B, n, L, synthetic.VOCAB + 1, seed=8)
    ties = {}

    def forward_ref(P):
        T.RELU_TIES = ties
        try:
            return T.forward_teacher(P, att, labels[..., :-1], am, h=8, n_enc=6, n_dec=6)
        finally:
            T.RELU_TIES = None
    _compare_model_with_oracle(model, forward_ref, lambda P: T.greedy(P, att, am, h=8, n_enc=6, n_dec=6, max_len=L), att, am, labels,
                               masks, relu_ties=ties)


@pytest.mark.parametrize('masked', [False, True])
def test_aoa_baseline_size_vs_oracle(masked):
    """configs/aoa/aoa.yml sizes (R=E=1024, h=8, 6 refiner layers), AoAModel.py:115-226; eval mode (the reference
    hard-codes 0.1 dropouts that no fixture can reproduce)"""
    from oracle import aoa as A
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    opt = synthetic.updown_opt(caption_model='aoa', input_encoding_size=1024, rnn_size=1024, att_hid_size=512, num_heads=8,
                               multi_head_scale=1, use_multi_head=2, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
                               mean_feats=1, ctx_drop=1, dropout_aoa=0.3, num_layers=2, drop_prob_lm=0.0, seq_length=6,
                               max_length=6)
    torch.manual_seed(12)
    model = models.setup(opt).to(DEV)
    model.eval()
    B, n, L, K = 2, 2, 6, 36
    _, att = shapes.feats(B, K=K, seed=6)
    am = None
    if masked:
        am = torch.ones(B, K)
        am[1, 20:] = 0
    labels, masks = _labels(B, n, L, synthetic.VOCAB + 1, seed=9)
    _compare_model_with_oracle(
        model, lambda P: A.forward_teacher(P, att, labels[..., :-1], am, h=8),
        lambda P: A.greedy(P, att, am, h=8, max_len=L), att, am, labels, masks)


def test_newfc_config_size_vs_oracle():
    """configs/fc.yml: R=E=512 (opts.py:44,50), V1=9488, 2048-d fc feats, bs10 (BASELINE configs[0]); NewFCModel
    AttModel.py:904-945 + LSTMCore FCModel.py:13-42"""
    from oracle import att_lstm as O
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    opt = synthetic.updown_opt(caption_model='newfc', input_encoding_size=512, rnn_size=512, drop_prob_lm=0.0,
                               seq_length=16, max_length=16)
    torch.manual_seed(13)
    model = models.setup(opt).to(DEV)
    model.train()
    B, n, L = 10, 5, 16
    fc, _ = shapes.feats(B, K=1, seed=7)
    labels, masks = _labels(B, n, L, synthetic.VOCAB + 1, seed=10)
    _compare_model_with_oracle(
        model, lambda P: O.newfc_forward_teacher(P, fc, labels[..., :-1]),
        lambda P: O.newfc_rollout_greedy(P, fc, max_len=L), None, None, labels, masks, fc=fc)


def test_new_self_critical_structure_loss_vs_reference_fixture():
    """StructureLosses(structure_loss_type='new_self_critical') (losses.py:168-187) on the log-probs / tokens the reference
    sampled, scores injected as in the fixture: the reference's ``nsc_loss``; and its gradient w.r.t. the log-probs against
    autograd of the oracle's restatement."""
    import argparse
    from oracle import att_lstm as O
    from imagecaptioning.pytorch_amd.captioning.modules import losses as L
    z = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    n = 2
    B = z['fc'].shape[0]
    sopt = argparse.Namespace(structure_loss_type='new_self_critical', train_sample_n=n, entropy_reward_weight=0,
                              self_cider_reward_weight=0, cider_reward_weight=1, bleu_reward_weight=0)
    scores = torch.from_numpy(z['nsc_scores'])
    saved = L.get_scores
    L.get_scores = lambda data_gts, gen_result, o, as_tensor=False: scores.to(DEV)
    try:
        slp = torch.from_numpy(z['sample_logp']).to(DEV).requires_grad_(True)
        seq = torch.from_numpy(z['sample_seq']).to(DEV)
        out = L.StructureLosses(sopt)(slp, seq, [None] * B)
        np.testing.assert_allclose(out['loss'].item(), z['nsc_loss'], rtol=1e-5)
        out['loss'].backward()
        for red in ('none',):
            rows = L.StructureLosses(sopt)(slp.detach(), seq, [None] * B, reduction=red)['loss']
            want_rows = O.new_self_critical_loss(torch.from_numpy(z['sample_logp']), torch.from_numpy(z['sample_seq']),
                                                 scores, n, reduction='none')
            np.testing.assert_allclose(rows.cpu().numpy(), want_rows.numpy(), rtol=1e-5, atol=1e-7)
    finally:
        L.get_scores = saved
    lp = torch.from_numpy(z['sample_logp']).requires_grad_(True)
    O.new_self_critical_loss(lp, torch.from_numpy(z['sample_seq']), scores, n).backward()
    np.testing.assert_allclose(slp.grad.cpu().numpy(), lp.grad.numpy(), rtol=1e-5, atol=1e-8)
    assert out['reward'].shape == (B, n)


def test_aoa_new_self_critical_step_gradients_vs_oracle():
    """BASELINE configs[4] (configs/aoa_nsc.yml): AoA sampled rollout (train_sample_n rows per image, injected Gumbel noise) ->
    StructureLosses 'new_self_critical' (losses.py:168-187, scores injected: CIDEr is pinned elsewhere) -> backward.  The oracle
    teacher-forces the sampled tokens (identical log-probs when dropout is off, as in the eval-mode fixture) and differentiates
    its own restatement of the loss: tokens are the model's, log-probs <= 1e-4, every gradient <= 1e-3 relative."""
    import argparse
    from oracle import aoa as A, att_lstm as O
    from test_model_api_gpu import tiny_opt
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules import losses as L
    z = np.load(os.path.join(GOLDEN, 'aoa_tiny.npz'))
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    opt = tiny_opt(caption_model='aoa', refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA', use_multi_head=2, num_heads=2,
                   multi_head_scale=1, mean_feats=1, ctx_drop=1, dropout_aoa=0.3, num_layers=2)
    model = models.setup(opt)
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')})
    model = model.to(DEV)
    model.eval()
    att, am = torch.from_numpy(u['att']), torch.from_numpy(u['att_masks'])
    B, n, Lmax, V1 = att.shape[0], 3, model.seq_length, model.vocab_size + 1
    N = B * n
    g = torch.Generator().manual_seed(17)
    gum = -torch.log(-torch.log(torch.rand(Lmax, N, V1, generator=g).clamp_min(1e-20)))
    scores = torch.rand(N, generator=g).double()
    seq, logp = model(None, att.to(DEV), am.to(DEV), opt={'sample_method': 'sample', 'sample_n': n, '_gumbel': gum.to(DEV)},
                      mode='sample')
    assert logp.requires_grad and seq.shape == (N, Lmax) and int((seq > 0).sum()) > N       # real captions, several lengths
    sopt = argparse.Namespace(structure_loss_type='new_self_critical', train_sample_n=n, entropy_reward_weight=0,
                              self_cider_reward_weight=0, cider_reward_weight=1, bleu_reward_weight=0)
    saved = L.get_scores
    L.get_scores = lambda data_gts, gen_result, o, as_tensor=False: scores.to(DEV)
    try:
        out = L.StructureLosses(sopt)(logp, seq, [None] * B)
    finally:
        L.get_scores = saved
    model.zero_grad()
    out['loss'].backward()
    # oracle: same weights, the sampled tokens teacher-forced
    P = {k: torch.from_numpy(z['P.' + k]).clone() for k in model.state_dict()}
    for v in P.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    seq_c = seq.cpu()
    inp = torch.cat([seq_c.new_zeros(N, 1), seq_c[:, :-1]], 1).view(B, n, Lmax)
    logp_o = A.forward_teacher(P, att, inp, am, h=2)
    live = torch.cat([seq_c.new_ones(N, 1), (seq_c[:, :-1] > 0).long()], 1).cumprod(1).bool()
    sel = logp.detach().cpu().gather(2, seq_c.unsqueeze(2)).squeeze(2)
    sel_o = logp_o.detach().gather(2, seq_c.unsqueeze(2)).squeeze(2)
    assert float(((sel - sel_o).abs() * live).max()) < 1e-4
    assert torch.equal(logp_o.detach()[:, 0].argmax(1) * 0, seq_c[:, 0] * 0)       # shapes line up
    loss_o = O.new_self_critical_loss(logp_o, seq_c, scores, n)
    assert abs(loss_o.item() - out['loss'].item()) < 1e-5
    loss_o.backward()
    floor = 1e-7 * max(float(p.grad.abs().max()) for p in P.values() if p.grad is not None)
    for k, p in model.named_parameters():
        ref = P[k].grad
        if k.endswith('linears.1.bias'):
            continue                                      # attention key bias: mathematically zero gradient
        assert float((p.grad.cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + floor, k


@pytest.mark.parametrize('R,E,A,F,K,V1,B,n,L,masked', [
    (52, 52, 20, 36, 7, 101, 3, 2, 6, False),        # multiples of 4 but of nothing larger; odd vocabulary and region count
    (50, 30, 18, 22, 5, 57, 2, 3, 5, True),          # nothing aligned: every vector path falls back to its scalar form
    (128, 64, 64, 100, 40, 1000, 9, 5, 8, True),     # K = 40: the upper limit of the register-resident attention kernels
    (64, 48, 32, 40, 41, 333, 4, 4, 7, False),       # K = 41: one region beyond it (the general kernels), 16 caption rows
    (256, 256, 512, 64, 36, 2048, 13, 5, 4, True),   # 65 caption rows: one more than a 64-row decode tile
])
def test_updown_odd_shapes_xe_gradients_and_greedy_tokens_vs_oracle(R, E, A, F, K, V1, B, n, L, masked):
    """Shape sweep of the UpDown hot path against the oracle run live: sizes that are not multiples of the vector widths / tile
    sizes, region counts on both sides of the fused attention kernels' register limits, and a row count one past the 64-row
    decode tile -- teacher-forced log-probs <= 1e-4, every XE gradient <= 1e-3 relative, greedy tokens exact (with and without
    att_masks)."""
    from oracle import att_lstm as O
    from test_model_api_gpu import tiny_opt
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion
    P = shapes.full_size_params(seed=R + K, V1=V1, R=R, E=E, A=A, F=F)
    g = torch.Generator().manual_seed(1000 + R)
    fc = (torch.randn(B, F, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, K, F, generator=g) * 0.5).clamp_min(0)
    am = None
    if masked:
        am = (torch.rand(B, K, generator=g) > 0.3).float()
        am[:, 0] = 1
        am[0] = 1                                                       # one image with every region valid
    labels, masks = _labels(B, n, L, V1, seed=R)
    opt = tiny_opt(vocab_size=V1 - 1, input_encoding_size=E, rnn_size=R, att_hid_size=A, fc_feat_size=F, att_feat_size=F,
                   seq_length=L, max_length=L, vocab={str(i): 'w%d' % i for i in range(1, V1)})
    model = models.setup(opt)
    model.load_state_dict(P)
    model = model.to(DEV)
    model.train()                                                       # drop_prob_lm = 0: deterministic
    d = lambda t: None if t is None else t.to(DEV)                      # noqa: E731
    logp = model(d(fc), d(att), d(labels[..., :-1]), d(am))
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = O.forward_teacher(Po, fc, att, labels[..., :-1], am)
    assert float((logp.detach().cpu() - ref.detach()).abs().max()) < 1e-4
    LanguageModelCriterion()(logp, d(labels[..., 1:]), d(masks[..., 1:])).backward()
    O.lm_criterion(ref, labels[..., 1:].reshape(B * n, -1), masks[..., 1:].reshape(B * n, -1)).backward()
    floor = 1e-7 * max(float(p.grad.abs().max()) for p in Po.values())
    for k, p in model.named_parameters():
        want = Po[k].grad
        if k == 'core.attention.alpha_net.bias':
            continue                                                    # mathematically zero (softmax shift invariance)
        assert float((p.grad.cpu() - want).abs().max()) <= 1e-3 * float(want.abs().max()) + floor, k
    model.eval()
    with torch.no_grad():
        seq, _ = model(d(fc), d(att), d(am), opt={'sample_method': 'greedy'}, mode='sample')
        want_seq, _ = O.rollout({k: v.detach() for k, v in Po.items()}, fc, att, am, method='greedy', max_len=L)
    assert torch.equal(seq.cpu(), want_seq)
    # sampled rollout (the SCST path: n rows per image, injected Gumbel noise) + RewardCriterion gradients
    from imagecaptioning.pytorch_amd.captioning.modules.losses import RewardCriterion
    N = B * n
    gum = -torch.log(-torch.log(torch.rand(L, N, V1, generator=g).clamp_min(1e-20)))
    reward = torch.randn(N, 1, generator=g).repeat(1, L)
    model.train()
    model.zero_grad()
    seq, slp = model(d(fc), d(att), d(am), opt={'sample_method': 'sample', 'sample_n': n, '_gumbel': d(gum)}, mode='sample')
    Ps = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    want_seq, want_lp = O.rollout(Ps, fc, att, am, method='sample', sample_n=n, gumbel=gum, max_len=L)
    assert torch.equal(seq.cpu(), want_seq)
    RewardCriterion()(slp, seq, d(reward)).backward()
    O.reward_criterion(want_lp, want_seq, reward).backward()
    floor = 1e-7 * max(float(p.grad.abs().max()) for p in Ps.values())
    for k, p in model.named_parameters():
        if k == 'core.attention.alpha_net.bias':
            continue
        want = Ps[k].grad
        assert float((p.grad.cpu() - want).abs().max()) <= 1e-3 * float(want.abs().max()) + floor, ('scst', k)


@pytest.mark.parametrize('d,h,dff,nl,K,V1,B,n,L,masked', [
    (48, 4, 100, 2, 7, 101, 3, 2, 5, False),         # head size 12, odd vocabulary / regions
    (40, 5, 52, 1, 41, 57, 2, 3, 6, True),           # head size 8, 41 regions, att_masks
    (96, 2, 36, 3, 36, 777, 4, 1, 9, True),          # head size 48, d_ff < d_model, one caption per image
])
def test_transformer_odd_shapes_vs_oracle(d, h, dff, nl, K, V1, B, n, L, masked):
    """Shape sweep of the Transformer path against oracle/transformer.py run live (sizes away from 512 / 2048 / 8 heads)."""
    from oracle import transformer as T
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    F = 44
    opt = synthetic.updown_opt(caption_model='transformer', input_encoding_size=d, rnn_size=dff, d_model=d, d_ff=dff, N_enc=nl,
                               N_dec=nl, num_att_heads=h, dropout=0.0, drop_prob_lm=0.0, seq_length=L, max_length=L,
                               vocab_size=V1 - 1, fc_feat_size=F, att_feat_size=F,
                               vocab={str(i): 'w%d' % i for i in range(1, V1)})
    torch.manual_seed(100 + d)
    model = models.setup(opt).to(DEV)
    model.train()
    _, att = shapes.feats(B, K=K, F=F, seed=d)
    am = None
    if masked:
        am = torch.ones(B, K)
        am[0, K // 2:] = 0
        am[-1, 1:] = 0                                 # an image with a single valid region
    labels, masks = _labels(B, n, L, V1, seed=d + 1)
    _compare_model_with_oracle(
        model, lambda P: T.forward_teacher(P, att, labels[..., :-1], am, h=h, n_enc=nl, n_dec=nl),
        lambda P: T.greedy(P, att, am, h=h, n_enc=nl, n_dec=nl, max_len=L), att, am, labels, masks)


@pytest.mark.parametrize('R,h,K,V1,B,n,L,masked', [
    (48, 4, 7, 101, 3, 2, 5, False),
    (64, 2, 41, 333, 2, 3, 6, True),
])
def test_aoa_odd_shapes_vs_oracle(R, h, K, V1, B, n, L, masked):
    """Shape sweep of the AoA path against oracle/aoa.py run live (eval mode, see test_aoa_baseline_size_vs_oracle)."""
    from oracle import aoa as A
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    F = 36
    opt = synthetic.updown_opt(caption_model='aoa', input_encoding_size=R, rnn_size=R, att_hid_size=R // 2, num_heads=h,
                               multi_head_scale=1, use_multi_head=2, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
                               mean_feats=1, ctx_drop=1, dropout_aoa=0.3, num_layers=2, drop_prob_lm=0.0, seq_length=L,
                               max_length=L, vocab_size=V1 - 1, fc_feat_size=F, att_feat_size=F,
                               vocab={str(i): 'w%d' % i for i in range(1, V1)})
    torch.manual_seed(200 + R)
    model = models.setup(opt).to(DEV)
    model.eval()
    _, att = shapes.feats(B, K=K, F=F, seed=R)
    am = None
    if masked:
        am = torch.ones(B, K)
        am[1, K // 3:] = 0
    labels, masks = _labels(B, n, L, V1, seed=R + 1)
    _compare_model_with_oracle(
        model, lambda P: A.forward_teacher(P, att, labels[..., :-1], am, h=h),
        lambda P: A.greedy(P, att, am, h=h, max_len=L), att, am, labels, masks)


@pytest.mark.parametrize('R,F,V1,B,n,L', [(50, 30, 57, 3, 2, 5), (132, 100, 1001, 13, 5, 7)])
def test_newfc_odd_shapes_vs_oracle(R, F, V1, B, n, L):
    """Shape sweep of the NewFC path against oracle/att_lstm.py run live (unaligned hidden size; 65 caption rows)."""
    from oracle import att_lstm as O
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    opt = synthetic.updown_opt(caption_model='newfc', input_encoding_size=R, rnn_size=R, drop_prob_lm=0.0, seq_length=L,
                               max_length=L, vocab_size=V1 - 1, fc_feat_size=F, att_feat_size=F,
                               vocab={str(i): 'w%d' % i for i in range(1, V1)})
    torch.manual_seed(300 + R)
    model = models.setup(opt).to(DEV)
    model.train()
    fc, _ = shapes.feats(B, K=1, F=F, seed=R)
    labels, masks = _labels(B, n, L, V1, seed=R + 2)
    _compare_model_with_oracle(
        model, lambda P: O.newfc_forward_teacher(P, fc, labels[..., :-1]),
        lambda P: O.newfc_rollout_greedy(P, fc, max_len=L), None, None, labels, masks, fc=fc)


def test_head_size_not_a_multiple_of_4_is_refused_loudly():
    """The one size restriction of the Transformer / AoA kernels (16-byte moves along the head dimension): a clear error, never a
    wrong result."""
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    opt = synthetic.updown_opt(caption_model='transformer', input_encoding_size=24, rnn_size=40, d_model=24, d_ff=40, N_enc=1,
                               N_dec=1, num_att_heads=4, dropout=0.0, drop_prob_lm=0.0, seq_length=5, max_length=5, vocab_size=56,
                               fc_feat_size=44, att_feat_size=44, vocab={str(i): 'w%d' % i for i in range(1, 57)})
    model = models.setup(opt).to(DEV)
    _, att = shapes.feats(2, K=7, F=44, seed=1)
    with pytest.raises(NotImplementedError, match='head size 6'):
        model(None, att.to(DEV), torch.randint(1, 57, (2, 1, 6), device=DEV), None)


@pytest.mark.parametrize('tag,family,B,seed,flat', [('t', 'transformer', 64, 11, False), ('t', 'transformer', 64, 11, True),
                                                    ('a', 'aoa', 10, 12, False), ('a', 'aoa', 10, 12, True)])
def test_transformer_and_aoa_at_the_baseline_batch_vs_the_reference_itself(tag, family, B, seed, flat):
    """the BASELINE *shapes* -- Transformer XE at bs64 x 5 captions, T=21 (configs[3]; 6 720 decoder rows,
    the DeferredGrads arena, every split-K plan of that size) and AoA at bs10 x 5, T=21 (the configs[4] batch) -- against
    outputs of the REAL reference (tests/golden/big_xe_grads.npz, `make_golden.py full2`): loss, target log-probs, three full
    distributions, every parameter gradient's norm and a 256-element probe."""
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion
    z = np.load(os.path.join(GOLDEN, 'big_xe_grads.npz'))
    model = models.setup(shapes.big_opt(family))
    model.load_state_dict(shapes.seeded_state({k: v.shape for k, v in model.state_dict().items()}, seed))
    model = model.to(DEV)
    if flat:                                  # the training layout: fused q | k | v and cross-attention K | V GEMMs (r4)
        model.flatten_parameters_()
    model.train() if family == 'transformer' else model.eval()
    fc, att = shapes.feats(B, seed=seed)
    labels, masks = shapes.c2_labels(B=B, seed=seed)
    labels, masks = labels.to(DEV), masks.to(DEV)
    logp = model(fc.to(DEV), att.to(DEV), labels[..., :-1], None)
    tgt = labels[..., 1:].reshape(-1, labels.shape[-1] - 1)
    got = logp.detach().gather(2, tgt[:, :logp.shape[1]].unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(got, z[tag + '_tgt_logp'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logp.detach()[0, :3].cpu().numpy(), z[tag + '_logp_row0'], rtol=0, atol=1e-4)
    loss = LanguageModelCriterion()(logp, labels[..., 1:], masks[..., 1:])
    assert abs(loss.item() - float(z[tag + '_loss'])) < 1e-4
    loss.backward()
    flat = getattr(model, '_flat', None)
    if flat is not None:
        flat.collect_grads()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
    # parameters the graph never reaches have an exact-zero reference gradient (norm 0): skip the relative check for them
    skip = tuple(k for k in grads if float(z['%s_gnorm.%s' % (tag, k)]) == 0.0 or k.endswith('alpha_net.bias'))
    check_grads_against_fixture(z, tag, grads, skip=skip)


@pytest.mark.parametrize('d,h,dff,nl,K,V1,B,n,L,F', [(48, 4, 100, 2, 7, 101, 3, 2, 6, 44),
                                                   (512, 8, 2048, 6, 36, 9488, 2, 2, 5, 2048)])      # r5: configs/transformer sizes
def test_transformer_scst_train_mode_differentiates_the_pass_it_sampled(d, h, dff, nl, K, V1, B, n, L, F):
    """Train-mode Transformer SCST (loss_wrapper.py:63-68): the KV-cached rollout and the teacher-forced pass
    that carries the gradient run under ONE dropout realisation.  With the masks of that realisation injected into
    oracle/transformer.py: (a) the log-probs the rollout sampled from == the differentiated log-probs == the oracle's
    (<= 2e-4), (b) every drawn token is arg-max(oracle log-prob + the injected Gumbel noise), i.e. the rollout is ON-policy
    for the differentiated pass, (c) the RewardCriterion gradient of every parameter matches the oracle's (<= 1e-3 relative)."""
    from oracle import transformer as T
    from imagecaptioning.pytorch_amd import synthetic, transformer_engine as E
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules import losses as Lm
    p_embed, p_drop = 0.3, 0.2
    opt = synthetic.updown_opt(caption_model='transformer', input_encoding_size=d, rnn_size=dff, d_model=d, d_ff=dff, N_enc=nl,
                               N_dec=nl, num_att_heads=h, dropout=p_drop, drop_prob_lm=p_embed, seq_length=L, max_length=L,
                               vocab_size=V1 - 1, fc_feat_size=F, att_feat_size=F,
                               vocab={str(i): 'w%d' % i for i in range(1, V1)})
    torch.manual_seed(4711)
    model = models.setup(opt).to(DEV)
    model.train()
    _, att = shapes.feats(B, K=K, F=F, seed=5)
    am = torch.ones(B, K)
    am[0, K // 2:] = 0
    N = B * n
    g = torch.Generator().manual_seed(23)
    gum = -torch.log(-torch.log(torch.rand(L, N, V1, generator=g).clamp_min(1e-20)))
    reward = torch.randn(N, L, generator=g)
    model._rng_calls = 0
    seq, logp = model(None, att.to(DEV), am.to(DEV), opt={'sample_method': 'sample', 'sample_n': n, '_gumbel': gum.to(DEV)},
                      mode='sample')
    assert logp.requires_grad and int((seq > 0).sum()) > N
    loss = Lm.RewardCriterion()(logp, seq, reward.to(DEV))
    model.zero_grad()
    loss.backward()

    # the realisation: the masks a TransformerGraph of the rollout's seed draws, in its order (encode, then decoder_masks)
    model._rng_calls = 0
    seed = model._next_seed()
    de = E.Dropper(p_embed, seed, torch.device(DEV), True)
    dd = E.Dropper(p_drop, seed ^ 0x5bd1e995, torch.device(DEV), True)
    named = {'att_embed': de(B * K, d).view(B, K, d)}
    for i in range(nl):
        named['enc%d.attn' % i] = dd(B, h, K, K)
        named['enc%d.res0' % i] = dd(B * K, d).view(B, K, d)
        named['enc%d.ff' % i] = dd(B * K, dff).view(B, K, dff)
        named['enc%d.res1' % i] = dd(B * K, d).view(B, K, d)
    named['tgt_embed'] = dd(N, L, d)
    for i in range(nl):
        named['dec%d.self.attn' % i] = dd(N, h, L, L)
        named['dec%d.res0' % i] = dd(N * L, d).view(N, L, d)
        named['dec%d.src.attn' % i] = dd(N, h, L, K)
        named['dec%d.res1' % i] = dd(N * L, d).view(N, L, d)
        named['dec%d.ff' % i] = dd(N * L, dff).view(N, L, dff)
        named['dec%d.res2' % i] = dd(N * L, d).view(N, L, d)
    named = {k: v.cpu() for k, v in named.items()}
    used = set()

    def drop(name, x):
        used.add(name)
        return x * named[name]

    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for v in P.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    P['model.tgt_embed.1.pe'] = model.model.tgt_embed[1].pe.cpu()
    seq_c = seq.cpu()
    inp = torch.cat([seq_c.new_zeros(N, 1), seq_c[:, :-1]], 1)
    T.RELU_TIES = ties = {}                 # hidden units whose ReLU input is within 1e-4 of zero for some token (see _compare_model_with_oracle)
    try:
        # the rollout came from _sample: the B images are encoded once and the memory repeated (TransformerModel.py:306-311)
        want = T.forward_teacher(P, att, inp, am, h=h, n_enc=nl, n_dec=nl, drop=drop, encode_per_caption=False)
    finally:
        T.RELU_TIES = None
    assert used == set(named)
    live = torch.cat([seq_c.new_ones(N, 1), (seq_c[:, :-1] > 0).long()], 1).cumprod(1).bool()
    # (a) differentiated log-probs == oracle under the same realisation, on the live steps
    got = logp.detach().cpu()
    assert float((got - want.detach())[live].abs().max()) <= 2e-4
    # (b) on-policy: each drawn token is the Gumbel-max of the SAME distribution
    for t in range(L):
        pick = (want.detach()[:, t] + gum[t]).argmax(1)
        assert torch.equal(seq_c[live[:, t], t], pick[live[:, t]]), t
    # (c) gradients
    mask = live.float()
    want_loss = -(want.gather(2, seq_c.unsqueeze(2)).squeeze(2) * reward * mask).sum() / mask.sum()
    assert abs(float(loss.detach()) - float(want_loss.detach())) <= 1e-4 * max(1.0, abs(float(want_loss.detach())))
    want_loss.backward()
    for k, prm in model.named_parameters():
        w = P[k].grad
        tie = ties.get(k.rsplit('.w_1.', 1)[0]) if (k.endswith('.w_1.weight') or k.endswith('.w_1.bias')) else None
        err = (prm.grad.cpu() - w).abs()
        if tie is not None and bool(tie.any()):         # a unit on the ReLU's kink takes one token's whole term either way
            err_rows = err.reshape(err.shape[0], -1).max(1)[0]
            assert float(err_rows[tie].max()) <= 5e-2 * float(w.abs().max()) + 1e-6, (k, 'rows with a ReLU tie')
            assert int(tie.sum()) < 0.05 * tie.numel(), (k, int(tie.sum()))
            err = err_rows[~tie]
        assert float(err.max()) <= 1e-3 * float(w.abs().max()) + 1e-6, k


@pytest.mark.parametrize('tag,seed,masked', [('u', 21, False), ('um', 22, True)])
def test_updown_xe_at_its_own_batch_bs64x5_vs_the_reference_itself(tag, seed, masked):
    """BASELINE configs[1] at ITS batch (VERDICT r3 missing #1(i)): bs64 x 5 captions = 320 rows, T = 21, R = E = 1000,
    V1 = 9488, +- att_masks -- 320 rows take the fat-GEMM decode path (gemm_x3), not the 64-row weight-streaming kernels the
    bs10 fixtures exercise.  Against outputs of the REAL reference (AttModel.py:126-164; tests/golden/updown_xe_bs64.npz,
    `make_golden.py full3`): loss and target log-probs <= 1e-4, three full distributions <= 1e-4, every parameter gradient's
    norm and 256-element probe <= 1e-3 relative."""
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion
    z = np.load(os.path.join(GOLDEN, 'updown_xe_bs64.npz'))
    model = models.setup(synthetic.updown_opt(drop_prob_lm=0.0))
    model.load_state_dict(shapes.full_size_params(seed=seed))
    model = model.to(DEV)
    model.train()
    B = 64
    fc, att = shapes.feats(B, seed=seed)
    am = shapes.ragged_masks(B, seed=seed).to(DEV) if masked else None
    labels, masks = shapes.c2_labels(B=B, seed=seed)
    labels, masks = labels.to(DEV), masks.to(DEV)
    logp = model(fc.to(DEV), att.to(DEV), labels[..., :-1], am)
    tgt = labels[..., 1:].reshape(-1, labels.shape[-1] - 1)
    got = logp.detach().gather(2, tgt[:, :logp.shape[1]].unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(got, z[tag + '_tgt_logp'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logp.detach()[0, :3].cpu().numpy(), z[tag + '_logp_row0'], rtol=0, atol=1e-4)
    loss = LanguageModelCriterion()(logp, labels[..., 1:], masks[..., 1:])
    assert abs(loss.item() - float(z[tag + '_loss'])) < 1e-4
    loss.backward()
    check_grads_against_fixture(z, tag, {k: p.grad for k, p in model.named_parameters()},
                                skip=('core.attention.alpha_net.bias',))
