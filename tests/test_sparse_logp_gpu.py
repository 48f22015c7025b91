"""The loss gradient travels from the criteria to the rollout backward in SPARSE form (selected-token gradient [N,L] + token ids
[+ row-sum gradient for label smoothing]) instead of a dense [N,L,V1] tensor (reference
losses.py:24, :81, :213, :258-262).  The results must equal the dense route exactly up to summation order."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_kernel_sparse_equals_dense_formula():
    import ctypes as C
    from imagecaptioning.pytorch_amd import _lib, ops
    from imagecaptioning.pytorch_amd._lib import ptr
    g = torch.Generator(device='cpu').manual_seed(0)
    N, L, T, V1 = 7, 6, 5, 9488
    logp = torch.log_softmax(torch.randn(N, L, V1, generator=g), 2).to(DEV)
    live = (torch.rand(N, L, generator=g) > 0.2).to(torch.uint8).to(DEV)
    tok = torch.randint(0, V1, (N, L), generator=g).to(DEV)
    a = torch.randn(N, L, generator=g).to(DEV)
    b = (torch.randn(N, L, generator=g) * 1e-3).to(DEV)
    dense_extra = (torch.randn(N, L, V1, generator=g) * 1e-3).to(DEV)
    for use_b, use_dense in ((False, False), (True, False), (True, True)):
        gd = torch.zeros(N, L, V1, device=DEV)
        gd.scatter_(2, tok.unsqueeze(2), a.unsqueeze(2))
        if use_b:
            gd += b.unsqueeze(2)
        if use_dense:
            gd += dense_extra
        want = torch.empty(T, N, V1, device=DEV)
        ops.logsoftmax_bwd(gd, None, logp, live, want, N, L, T, V1)
        sp = _lib.SparseLogpGrad()
        sp.g_sel, sp.tok, sp.tok_ld = ptr(a), ptr(tok), L
        sp.g_sum = ptr(b) if use_b else None
        got = torch.full((T, N, V1), float('nan'), device=DEV)
        ops.logsoftmax_bwd(dense_extra if use_dense else None, sp, logp, live, got, N, L, T, V1)
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9
        # against the definition, in float64
        p = logp.double().exp()
        ref = gd.double() - p * gd.double().sum(2, keepdim=True)
        ref = ref * live.unsqueeze(2).double()
        assert float((got.double() - ref[:, :T].transpose(0, 1)).abs().max()) < 1e-6


def _grads(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def _check_same(ga, gb):
    floor = 1e-7 * max(float(v.abs().max()) for v in gb.values())      # gradients that are mathematically zero hold rounding noise
    for k in ga:
        scale = float(gb[k].abs().max())
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-5 * scale + floor, k


@pytest.mark.parametrize('family', ['updown', 'newfc', 'transformer', 'aoa'])
@pytest.mark.parametrize('crit_name', ['xe', 'ls'])
def test_xe_and_label_smoothing_sparse_route_equals_dense_route(family, crit_name):
    from test_model_api_gpu import tiny_opt
    from imagecaptioning.pytorch_amd import sparse_logp
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion, LabelSmoothing
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    z = np.load(os.path.join(GOLDEN, family + '_tiny.npz'))
    kw = {'updown': {}, 'newfc': dict(caption_model='newfc'),
          'transformer': dict(caption_model='transformer', N_enc=2, N_dec=2, d_model=16, d_ff=32, num_att_heads=2, dropout=0.0),
          'aoa': dict(caption_model='aoa', refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA', use_multi_head=2, num_heads=2,
                      multi_head_scale=1, mean_feats=1, ctx_drop=1, dropout_aoa=0.3, num_layers=2)}[family]
    model = models.setup(tiny_opt(**kw))
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')})
    model = model.to(DEV)
    model.eval() if family == 'aoa' else model.train()
    fc, att, am = (torch.from_numpy(u[k]).to(DEV) for k in ('fc', 'att', 'att_masks'))
    labels, masks = torch.from_numpy(u['labels']).to(DEV), torch.from_numpy(u['masks']).to(DEV)
    crit = LanguageModelCriterion() if crit_name == 'xe' else LabelSmoothing(smoothing=0.2)
    taken = []
    orig = sparse_logp.split_grad

    def spy(g_logp, sink, like=None):
        out = orig(g_logp, sink, like)
        taken.append((out[0] is None, out[1] is not None))
        return out
    sparse_logp.split_grad = spy
    try:
        model.zero_grad()
        logp = model(fc, att, labels[..., :-1], am)
        assert getattr(logp, '_capmi_sink', None) is not None
        loss_s = crit(logp, labels[..., 1:], masks[..., 1:])
        loss_s.backward()
        g_sparse = _grads(model)
        assert taken == [(True, True)], taken            # no dense gradient reached the rollout, a sparse one did
        model.zero_grad()
        logp = model(fc, att, labels[..., :-1], am) * 1.0       # a plain tensor: the dense route (gather + autograd scatter)
        loss_d = crit(logp, labels[..., 1:], masks[..., 1:])
        loss_d.backward()
        assert taken[-1] == (False, False)
    finally:
        sparse_logp.split_grad = orig
    assert abs(loss_s.item() - loss_d.item()) < 1e-6
    _check_same(g_sparse, _grads(model))


def test_scst_reward_criterion_sparse_route_and_mixed_dense_use():
    """RewardCriterion on a sampled rollout (the SCST step) + an entropy-like term that reads the dense tensor: the sparse and
    the dense gradient parts are added in the rollout backward."""
    from test_model_api_gpu import golden_model
    from imagecaptioning.pytorch_amd.captioning.modules.losses import RewardCriterion
    z, model = golden_model(True)
    model.train()
    fc, att, am = (torch.from_numpy(z[k]).to(DEV) for k in ('fc', 'att', 'att_masks'))
    forced = torch.from_numpy(z['sample_seq']).to(DEV)
    reward = torch.from_numpy(z['sample_reward']).to(DEV)

    def run(dense_route, with_entropy):
        model.zero_grad()
        seq, logp = model(fc, att, am, opt={'sample_method': 'sample', 'sample_n': 2, 'temperature': 1.3, '_forced_seq': forced},
                          mode='sample')
        lp = logp * 1.0 if dense_route else logp
        loss = RewardCriterion()(lp, seq, reward)
        if with_entropy:
            loss = loss + 1e-3 * (logp.exp() * logp).sum(2).mean()
        loss.backward()
        return loss.item(), _grads(model)
    l0, g0 = run(False, False)
    np.testing.assert_allclose(l0, z['rl_loss'], rtol=1e-5)
    for k, p in g0.items():
        ref = z['rl_grad.' + k]
        np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=5e-4, atol=1e-6 + 2e-5 * np.abs(ref).max(), err_msg=k)
    l1, g1 = run(True, False)
    _check_same(g0, g1)
    l2, g2 = run(False, True)
    l3, g3 = run(True, True)
    assert abs(l2 - l3) < 1e-6
    _check_same(g2, g3)


def test_transformer_scst_masked_logprobs_keep_the_sparse_route():
    from test_model_api_gpu import tiny_opt
    from imagecaptioning.pytorch_amd import sparse_logp
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.losses import RewardCriterion
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    z = np.load(os.path.join(GOLDEN, 'transformer_tiny.npz'))
    model = models.setup(tiny_opt(caption_model='transformer', N_enc=2, N_dec=2, d_model=16, d_ff=32, num_att_heads=2, dropout=0.0))
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')})
    model = model.to(DEV)
    model.train()
    att = torch.from_numpy(u['att']).to(DEV)
    torch.manual_seed(3)
    model._rng_calls = 5
    seq, logp = model(None, att, None, opt={'sample_method': 'sample', 'sample_n': 2}, mode='sample')
    assert getattr(logp, '_capmi_masked', None) is not None
    reward = torch.randn(seq.shape[0], 1, device=DEV).expand(-1, seq.shape[1])
    taken = []
    orig = sparse_logp.split_grad

    def spy(g_logp, sink, like=None):
        out = orig(g_logp, sink, like)
        taken.append((out[0] is None, out[1] is not None))
        return out
    sparse_logp.split_grad = spy
    try:
        model.zero_grad()
        loss = RewardCriterion()(logp, seq, reward)
        loss.backward()
        gs = _grads(model)
        assert taken == [(True, True)]
        model.zero_grad()
        model._rng_calls = 5
        seq2, logp2 = model(None, att, None, opt={'sample_method': 'sample', 'sample_n': 2}, mode='sample')
        assert torch.equal(seq, seq2)
        RewardCriterion()(logp2 * 1.0, seq2, reward).backward()
    finally:
        sparse_logp.split_grad = orig
    _check_same(gs, _grads(model))


def test_fused_scst_rollout_keeps_the_sparse_route():
    """LossWrapper's SCST branch on the fused greedy+sample rollout (AttModel.scst_rollouts returns the sampled rows as a slice
    of the rollout's tensor): the RewardCriterion gradient must still arrive sparse, and equal the dense route."""
    from test_model_api_gpu import golden_model
    from imagecaptioning.pytorch_amd import sparse_logp
    from imagecaptioning.pytorch_amd.captioning.modules.losses import RewardCriterion
    z, model = golden_model(True)
    model.train()
    fc, att, am = (torch.from_numpy(z[k]).to(DEV) for k in ('fc', 'att', 'att_masks'))
    B, n, L, V1 = fc.shape[0], 2, model.seq_length, model.vocab_size + 1
    g = torch.Generator().manual_seed(5)
    gum = -torch.log(-torch.log(torch.rand(L, B * n + B, V1, generator=g).clamp_min(1e-20))).to(DEV)
    reward = torch.randn(B * n, 1, generator=g).expand(-1, L).to(DEV)
    taken = []
    orig = sparse_logp.split_grad

    def spy(g_logp, sink, like=None):
        out = orig(g_logp, sink, like)
        taken.append((out[0] is None, out[1] is not None))
        return out
    sparse_logp.split_grad = spy
    try:
        res = []
        for dense in (False, True):
            model.zero_grad()
            model._rng_calls = 9
            greedy, gen, logp = model.scst_rollouts(fc, att, am, sample_n=n, _gumbel=gum)
            assert getattr(logp, '_capmi_rows', None) is not None
            RewardCriterion()(logp * 1.0 if dense else logp, gen, reward).backward()
            res.append((gen.clone(), _grads(model)))
    finally:
        sparse_logp.split_grad = orig
    assert taken == [(True, True), (False, False)], taken
    assert torch.equal(res[0][0], res[1][0])
    _check_same(res[0][1], res[1][1])


def test_fused_reward_criterion_equals_generic_route_and_launch_count_drops():
    """The SCST branch of LossWrapper end to end: RewardCriterion served by capmi_reward_criterion from the rollout's saved
    selected log-probs (loss and every gradient equal to the generic gather route), hypotheses scored in place, masks in
    one launch."""
    from test_model_api_gpu import golden_model
    from imagecaptioning.pytorch_amd import sparse_logp
    from imagecaptioning.pytorch_amd.captioning.modules.losses import RewardCriterion
    z, model = golden_model(True)
    model.train()
    fc, att, am = (torch.from_numpy(z[k]).to(DEV) for k in ('fc', 'att', 'att_masks'))
    B, n, L, V1 = fc.shape[0], 2, model.seq_length, model.vocab_size + 1
    g = torch.Generator().manual_seed(6)
    gum = -torch.log(-torch.log(torch.rand(L, B * n + B, V1, generator=g).clamp_min(1e-20))).to(DEV)
    adv = torch.randn(B * n, generator=g).to(DEV)
    res = []
    for mode in ('fused', 'generic', 'fused_rows'):
        model.zero_grad()
        model._rng_calls = 11
        greedy, gen, logp = model.scst_rollouts(fc, att, am, sample_n=n, _gumbel=gum)
        assert gen._capmi_all.shape[0] == B * n + B and gen.data_ptr() == gen._capmi_all.data_ptr()
        reward = adv.unsqueeze(1).expand(-1, L)
        if mode == 'generic':
            saved, sparse_logp._DENSE = sparse_logp._DENSE, False
            fr = sparse_logp.fused_reward_criterion
            sparse_logp.fused_reward_criterion = lambda *a, **k: None
            import imagecaptioning.pytorch_amd.captioning.modules.losses as Lm
            keep = Lm.fused_reward_criterion
            Lm.fused_reward_criterion = lambda *a, **k: None
            try:
                loss = RewardCriterion()(logp, gen.data, reward)
            finally:
                Lm.fused_reward_criterion = keep
                sparse_logp.fused_reward_criterion = fr
                sparse_logp._DENSE = saved
        elif mode == 'fused':
            loss = RewardCriterion()(logp, gen.data, reward)
            assert loss.grad_fn is not None and 'FusedReward' in type(loss.grad_fn).__name__
        else:
            rows = RewardCriterion()(logp, gen.data, reward, reduction='none')
            assert rows.shape == (B * n,)
            m = torch.cat([gen.new_ones(B * n, 1), (gen[:, :-1] > 0).long()], 1).float()
            loss = (rows * m.sum(1)).sum() / m.sum()              # the mean form rebuilt from the per-row losses
        loss.backward()
        res.append((loss.item(), _grads(model), gen.clone()))
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][2], res[2][2])
    assert abs(res[0][0] - res[1][0]) < 1e-6 and abs(res[0][0] - res[2][0]) < 1e-5
    _check_same(res[0][1], res[1][1])
    _check_same(res[2][1], res[1][1])


def test_dropout_masks_one_launch_equals_four_and_eval_rows():
    from imagecaptioning.pytorch_amd import ops
    seed, p = 1234567, 0.5
    shapes = [((3, 16), 0), ((3, 5, 16), 1 << 40), ((4, 9, 12), 2 << 40), ((4, 9, 16), 3 << 40)]
    one = ops.dropout_masks([(s, off, None, DEV) for s, off in shapes], p, seed)
    for m, (s, off) in zip(one, shapes):
        assert torch.equal(m, ops.dropout_mask(s, p, seed, off, DEV))
        assert set(m.unique().tolist()) <= {0.0, 2.0}
    ev = ops.dropout_masks([(shapes[2][0], shapes[2][1], 6, DEV)], p, seed)[0]
    assert torch.equal(ev[:, :6], one[2][:, :6]) and bool((ev[:, 6:] == 1).all())


@pytest.mark.parametrize('loss_type,ew', [('seqnll', 0.0), ('softmax_margin', 0.0), ('risk', 0.2), ('best_of_n', 0.0),
                                          ('new_self_critical', 0.2)])
def test_structure_loss_types_sparse_route_equals_dense_route(loss_type, ew):
    """Every log-probability StructureLosses type (losses.py:40-202; values pinned to the reference class on CPU by
    tests/test_structure_losses.py) on a sampled UpDown rollout: the loss only reads the sampled tokens' log-probabilities, so
    its gradient reaches the rollout backward in sparse form -- same loss and parameter gradients as the dense autograd route
    (the entropy reward reads the dense rows without gradient)."""
    import argparse
    from test_model_api_gpu import golden_model
    from imagecaptioning.pytorch_amd.captioning.modules import losses as L
    z, model = golden_model(True)
    model.train()
    fc, att, am = (torch.from_numpy(z[k]).to(DEV) for k in ('fc', 'att', 'att_masks'))
    forced = torch.from_numpy(z['sample_seq']).to(DEV)
    N = forced.shape[0]
    scores = torch.linspace(0.1, 0.9, N, dtype=torch.float64).flip(0).to(DEV) ** 2
    opt = argparse.Namespace(structure_loss_type=loss_type, train_sample_n=2, entropy_reward_weight=ew, self_cider_reward_weight=0)
    saved = L.get_scores
    L.get_scores = lambda data_gts, gen_result, o, as_tensor=False: scores.clone()

    def run(dense_route):
        model.zero_grad()
        seq, logp = model(fc, att, am, opt={'sample_method': 'sample', 'sample_n': 2, '_forced_seq': forced}, mode='sample')
        out = L.StructureLosses(opt)(logp * 1.0 if dense_route else logp, seq, [None] * (N // 2))
        out['loss'].backward()
        return out['loss'].item(), _grads(model)
    try:
        l0, g0 = run(False)
        l1, g1 = run(True)
    finally:
        L.get_scores = saved
    assert np.isfinite(l0) and abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
    _check_same(g0, g1)
