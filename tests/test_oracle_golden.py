"""Pins oracle/att_lstm.py to the REAL reference through tests/golden/*.npz (made by
tests/golden/make_golden.py, which imports /root/reference).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import att_lstm as O
from conftest import GOLDEN

TOL = dict(rtol=1e-5, atol=2e-6)


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')}
    return z, P


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_updown_teacher_forced_logprobs_loss_and_grads(tag):
    z, P = load('updown_tiny.npz')
    for v in P.values():
        v.requires_grad_(True)
    fc, att = torch.from_numpy(z['fc']), torch.from_numpy(z['att'])
    labels, masks = torch.from_numpy(z['labels']), torch.from_numpy(z['masks'])
    am = torch.from_numpy(z['att_masks']) if tag == 'mask' else None
    logp = O.forward_teacher(P, fc, att, labels[..., :-1], am)
    np.testing.assert_allclose(logp.detach().numpy(), z['xe_logp_' + tag], **TOL)
    # the early-break columns must be exactly zero like the reference's
    assert (z['xe_logp_' + tag][:, -1] == 0).all() and (logp.detach().numpy()[:, -1] == 0).all()
    loss = O.lm_criterion(logp, labels[..., 1:], masks[..., 1:])
    np.testing.assert_allclose(loss.item(), z['xe_loss_' + tag], rtol=1e-6)
    rows = O.lm_criterion(logp, labels[..., 1:], masks[..., 1:], reduction='none')
    np.testing.assert_allclose(rows.detach().numpy(), z['xe_loss_rows_' + tag], rtol=1e-5)
    ls = O.label_smoothing_criterion(logp, labels[..., 1:], masks[..., 1:], 0.2)
    np.testing.assert_allclose(ls.item(), z['ls_loss_' + tag], rtol=1e-5)
    loss.backward()
    for k, p in P.items():
        np.testing.assert_allclose(p.grad.numpy(), z['xe_grad_%s.%s' % (tag, k)], rtol=2e-4, atol=2e-7, err_msg=k)


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_updown_greedy_token_exact(tag):
    z, P = load('updown_tiny.npz')
    fc, att = torch.from_numpy(z['fc']), torch.from_numpy(z['att'])
    am = torch.from_numpy(z['att_masks']) if tag == 'mask' else None
    with torch.no_grad():
        seq, slp = O.rollout(P, fc, att, am, method='greedy', max_len=8)
    assert np.array_equal(seq.numpy(), z['greedy_seq_' + tag])
    np.testing.assert_allclose(slp.numpy(), z['greedy_logp_' + tag], **TOL)


def test_updown_sampled_rollout_forced_tokens_and_reward_grads():
    """The reference sampled `sample_seq` with torch's generator; teacher-forcing those tokens
    through the oracle must reproduce its dense log-probs (un-tempered although temperature 1.3
    was used to sample), the RewardCriterion loss and every parameter gradient."""
    z, P = load('updown_tiny.npz')
    for v in P.values():
        v.requires_grad_(True)
    fc, att, am = (torch.from_numpy(z[k]) for k in ('fc', 'att', 'att_masks'))
    forced = torch.from_numpy(z['sample_seq'])
    seq, slp = O.rollout(P, fc, att, am, method='sample', sample_n=2, temperature=1.3, max_len=8, forced=forced)
    assert np.array_equal(seq.numpy(), z['sample_seq'])
    np.testing.assert_allclose(slp.detach().numpy(), z['sample_logp'], **TOL)
    loss = O.reward_criterion(slp, seq, torch.from_numpy(z['sample_reward']))
    np.testing.assert_allclose(loss.item(), z['rl_loss'], rtol=1e-5)
    loss.backward()
    for k, p in P.items():
        np.testing.assert_allclose(p.grad.numpy(), z['rl_grad.' + k], rtol=2e-4, atol=2e-7, err_msg=k)
    nsc = O.new_self_critical_loss(slp.detach(), seq, torch.from_numpy(z['nsc_scores']), 2)
    np.testing.assert_allclose(nsc.item(), z['nsc_loss'], rtol=1e-5)


def test_gumbel_sampling_is_categorical():
    """Gumbel-max over logp/T is the distribution Categorical(logits=logp/T) (CaptionModel.py:405)."""
    g = torch.Generator().manual_seed(0)
    logp = torch.log_softmax(torch.randn(1, 6, generator=g), 1)
    T = 0.7
    u = torch.rand(200000, 6, generator=g).clamp_min(1e-20)
    it = torch.max(logp / T - torch.log(-torch.log(u)), 1)[1]
    freq = torch.bincount(it, minlength=6).float() / it.numel()
    np.testing.assert_allclose(freq.numpy(), torch.softmax(logp / T, 1)[0].numpy(), atol=5e-3)


def test_newfc_teacher_forced_and_greedy():
    z, P = load('newfc_tiny.npz')
    for v in P.values():
        v.requires_grad_(True)
    fc = torch.from_numpy(z['fc'])
    labels, masks = torch.from_numpy(z['labels']), torch.from_numpy(z['masks'])
    logp = O.newfc_forward_teacher(P, fc, labels[..., :-1])
    np.testing.assert_allclose(logp.detach().numpy(), z['xe_logp'], **TOL)
    loss = O.lm_criterion(logp, labels[..., 1:], masks[..., 1:])
    np.testing.assert_allclose(loss.item(), z['xe_loss'], rtol=1e-6)
    loss.backward()
    for k, p in P.items():
        np.testing.assert_allclose(p.grad.numpy(), z['xe_grad.' + k], rtol=2e-4, atol=2e-7, err_msg=k)


def test_newfc_greedy_token_exact():
    z, P = load('newfc_tiny.npz')
    with torch.no_grad():
        seq, slp = O.newfc_rollout_greedy(P, torch.from_numpy(z['fc']), max_len=8)
    assert np.array_equal(seq.numpy(), z['greedy_seq'])
    np.testing.assert_allclose(slp.numpy(), z['greedy_logp'], **TOL)


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_transformer_teacher_forced_loss_grads_and_greedy(tag):
    from oracle import transformer as T
    z = np.load(os.path.join(GOLDEN, 'transformer_tiny.npz'))
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith('.pe'):
            v.requires_grad_(True)
    att = torch.from_numpy(u['att'])
    am = torch.from_numpy(u['att_masks']) if tag == 'mask' else None
    labels, masks = torch.from_numpy(u['labels']), torch.from_numpy(u['masks'])
    logp = T.forward_teacher(P, att, labels[..., :-1], am, h=2, n_enc=2, n_dec=2)
    np.testing.assert_allclose(logp.detach().numpy(), z['xe_logp_' + tag], rtol=1e-5, atol=3e-6)
    loss = O.lm_criterion(logp, labels[..., 1:], masks[..., 1:])
    np.testing.assert_allclose(loss.item(), z['xe_loss_' + tag], rtol=1e-6)
    loss.backward()
    for k, p in P.items():
        if p.requires_grad:
            np.testing.assert_allclose(p.grad.numpy(), z['xe_grad_%s.%s' % (tag, k)], rtol=3e-4, atol=3e-7, err_msg=k)
    with torch.no_grad():
        seq, slp = T.greedy(P, att, am, h=2, n_enc=2, n_dec=2, max_len=8)
    assert np.array_equal(seq.numpy(), z['greedy_seq_' + tag])
    np.testing.assert_allclose(slp.numpy(), z['greedy_logp_' + tag], rtol=1e-5, atol=3e-6)


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_aoa_teacher_forced_loss_grads_and_greedy(tag):
    from oracle import aoa as A
    z = np.load(os.path.join(GOLDEN, 'aoa_tiny.npz'))
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]).requires_grad_(True) for k in z.files if k.startswith('P.')}
    att = torch.from_numpy(u['att'])
    am = torch.from_numpy(u['att_masks']) if tag == 'mask' else None
    labels, masks = torch.from_numpy(u['labels']), torch.from_numpy(u['masks'])
    logp = A.forward_teacher(P, att, labels[..., :-1], am, h=2)
    np.testing.assert_allclose(logp.detach().numpy(), z['xe_logp_' + tag], rtol=1e-5, atol=3e-6)
    loss = O.lm_criterion(logp, labels[..., 1:], masks[..., 1:])
    np.testing.assert_allclose(loss.item(), z['xe_loss_' + tag], rtol=1e-6)
    loss.backward()
    for k, p in P.items():
        np.testing.assert_allclose(p.grad.numpy(), z['xe_grad_%s.%s' % (tag, k)], rtol=3e-4, atol=3e-7, err_msg=k)
    with torch.no_grad():
        seq, slp = A.greedy(P, att, am, h=2, max_len=8)
    assert np.array_equal(seq.numpy(), z['greedy_seq_' + tag])
    np.testing.assert_allclose(slp.numpy(), z['greedy_logp_' + tag], rtol=1e-5, atol=3e-6)


@pytest.mark.parametrize('tag', ['p50', 'p100'])
def test_updown_scheduled_sampling_replays_the_reference(tag):
    """AttModel._forward with ss_prob > 0 (AttModel.py:145-154): after the same torch.manual_seed the oracle makes the same
    uniform_/multinomial calls as the reference, so log-probs, loss and every gradient must coincide."""
    z, P = load('updown_tiny.npz')
    g = np.load(os.path.join(GOLDEN, 'updown_tiny_ss.npz'))
    for v in P.values():
        v.requires_grad_(True)
    fc, att, am = (torch.from_numpy(z[k]) for k in ('fc', 'att', 'att_masks'))
    labels, masks = torch.from_numpy(z['labels']), torch.from_numpy(z['masks'])
    torch.manual_seed(int(g[tag + '_seed']))
    logp = O.forward_teacher(P, fc, att, labels[..., :-1], am, ss_prob=float(g[tag + '_prob']))
    np.testing.assert_allclose(logp.detach().numpy(), g[tag + '_logp'], **TOL)
    assert np.abs(g[tag + '_logp'] - z['xe_logp_mask']).max() > 1e-2          # the sampled inputs really changed the outputs
    loss = O.lm_criterion(logp, labels[..., 1:], masks[..., 1:])
    np.testing.assert_allclose(loss.item(), g[tag + '_loss'], rtol=1e-6)
    loss.backward()
    for k, p in P.items():
        np.testing.assert_allclose(p.grad.numpy(), g['%s_grad.%s' % (tag, k)], rtol=2e-4, atol=2e-7, err_msg=k)


def test_aoa_sampled_rollout_oracle_is_consistent_with_teacher_forcing():
    """oracle/aoa.sample (r5: the CPU baseline of bench.py --config aoa_nsc times it) restates AttModel._sample for sample_n rows
    per image WITH the autograd graph.  Teacher-forcing the tokens it drew (oracle forward_teacher, pinned by aoa_tiny.npz) must
    give the same log-probs on the live steps, rows after their end are zero, and the new_self_critical loss differentiates."""
    import numpy as np
    import torch
    from oracle import aoa as A, att_lstm as O
    z = np.load(os.path.join(GOLDEN, 'aoa_tiny.npz'))
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(z[k].dtype.kind == 'f') for k in z.files if k.startswith('P.')}
    att, am = torch.from_numpy(u['att']), torch.from_numpy(u['att_masks'])
    B, n, L = att.shape[0], 3, 8
    seq, slp = A.sample(P, att, am, 2, L, n=n, gen=torch.Generator().manual_seed(11))
    assert seq.shape == (B * n, L) and slp.shape[:2] == (B * n, L) and slp.requires_grad
    inp = torch.cat([seq.new_zeros(B * n, 1), seq[:, :-1]], 1).view(B, n, L)
    want = A.forward_teacher(P, att, inp, am, h=2)
    live = torch.cat([seq.new_ones(B * n, 1), (seq[:, :-1] > 0).long()], 1).cumprod(1).bool()
    assert float((slp - want)[live].abs().max()) < 1e-5
    assert float(slp[~live].abs().max() if (~live).any() else 0.0) == 0.0
    assert bool((seq[~live] == 0).all())
    loss = O.new_self_critical_loss(slp, seq, torch.rand(B * n, generator=torch.Generator().manual_seed(1)).double(), n)
    loss.backward()
    assert float(P['logit.weight'].grad.abs().sum()) > 0 and float(P['refiner.layers.0.self_attn.linears.0.weight'].grad.abs().sum()) > 0


@pytest.mark.parametrize('family', ['newfc', 'transformer', 'aoa'])
def test_raw_logit_mode_of_the_oracles_is_the_reference_logprobs_before_the_softmax(family):
    """`want_logsoftmax=False` (AttModel.get_logprobs_state(output_logsoftmax=0), AttModel.py:171-175: `logprobs = self.logit(output)`)
    of the three restatements that gained it in r5: log_softmax of the returned rows == the REFERENCE's teacher-forced log-probs of
    the fixtures, the rows themselves are not normalised, and the switch changes nothing else (same zero columns)."""
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    if family == 'newfc':
        z, P = load('newfc_tiny.npz')
        labels = torch.from_numpy(z['labels'])
        raw = O.newfc_forward_teacher(P, torch.from_numpy(z['fc']), labels[..., :-1], want_logsoftmax=False)
        want = z['xe_logp']
    else:
        z = np.load(os.path.join(GOLDEN, family + '_tiny.npz'))
        P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')}
        labels, att = torch.from_numpy(u['labels']), torch.from_numpy(u['att'])
        if family == 'transformer':
            from oracle import transformer as T
            raw = T.forward_teacher(P, att, labels[..., :-1], None, h=2, n_enc=2, n_dec=2, want_logsoftmax=False)
        else:
            from oracle import aoa as A
            raw = A.forward_teacher(P, att, labels[..., :-1], None, h=2, want_logsoftmax=False)
        want = z['xe_logp_nomask']
    with torch.no_grad():
        written = torch.from_numpy(np.abs(want).sum(-1) > 0)                 # steps the reference ran (AttModel.py:140-143 breaks early)
        lp = torch.log_softmax(raw, -1) * written.unsqueeze(-1)
        np.testing.assert_allclose(lp.numpy(), want, rtol=1e-5, atol=3e-6)
        assert float((torch.logsumexp(raw, -1)[written]).abs().max()) > 1e-3   # logits, not log-probabilities
        if family != 'transformer':                                          # (TransformerModel._forward has no early break)
            assert float(raw[~written].abs().max() if bool((~written).any()) else 0.0) == 0.0


# ------------------------------------------------------------------------------------------------------------------------
# TRAIN MODE: tests/golden/train_mode.npz is the REFERENCE in train() with drop_prob_lm 0.5 / Transformer dropout 0.1 /
# dropout_aoa 0.3, every F.dropout call recorded in call order (make_golden.py `train`, DropRecorder).  The oracles replay
# the recorded masks through their drop hooks: which tensor gets which mask -- the thing the other fixtures (all at rate 0 or
# in eval()) cannot see -- is pinned to the reference here, not to a reading of it.
# ------------------------------------------------------------------------------------------------------------------------

class _Masks:
    """The recorded masks of one reference run, handed out in call order, pre-scaled (keep / (1 - p)) like F.dropout."""

    def __init__(self, z, key):
        self.p = z[key + '.drop_p']
        self.m = []
        for i, p in enumerate(self.p):
            shape = tuple(z['%s.drop%03d.shape' % (key, i)])
            bits = np.unpackbits(z['%s.drop%03d' % (key, i)])[:int(np.prod(shape))].reshape(shape)
            self.m.append(torch.from_numpy(bits.astype(np.float32)) / (1.0 - float(p)))
        self.i = 0

    def take(self, shape=None, p=None):
        assert self.i < len(self.m), 'the oracle drops a tensor the reference does not (call %d)' % self.i
        m = self.m[self.i]
        if shape is not None:
            assert tuple(m.shape) == tuple(shape), 'dropout call %d: the reference masked %s, the oracle %s' % (self.i, tuple(m.shape), tuple(shape))
        if p is not None:
            assert abs(float(self.p[self.i]) - p) < 1e-6, 'dropout call %d: rate %g in the reference, %g expected' % (self.i, self.p[self.i], p)
        self.i += 1
        return m

    def left(self):
        return len(self.m) - self.i


def _unpack_regions(packed, am):
    """The reference applies att_embed (and its Dropout) to the PACKED regions (AttModel.pack_wrapper, AttModel.py:44-49:
    rows sorted by length descending, then time-major).  Scatter the packed mask [sum(len), D] back to [B, K', D] (K' = longest
    row); padded positions get 0 -- they are zeroed by the region mask anyway."""
    lens = am.long().sum(1)
    order = sorted(range(len(lens)), key=lambda b: -int(lens[b]))
    assert len(set(lens.tolist())) == len(lens), 'fixture rows must have distinct lengths (ties would need torch.sort order)'
    out = packed.new_zeros(len(lens), int(lens.max()), packed.shape[1])
    r = 0
    for t in range(int(lens.max())):
        for b in order:
            if int(lens[b]) > t:
                out[b, t] = packed[r]
                r += 1
    assert r == packed.shape[0]
    return out


def _hook(masks, am, rates):
    """drop(name, x) for oracle/transformer.py and oracle/aoa.py: next recorded mask, shape- and rate-checked."""
    def drop(name, x):
        p = next(v for k, v in rates if name.startswith(k) or name.endswith(k))
        if name == 'att_embed' and am is not None:
            return x * _unpack_regions(masks.take(None, p), am)
        return x * masks.take(x.shape, p)
    return drop


def _train_inputs():
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    g = np.load(os.path.join(GOLDEN, 'train_mode.npz'))
    t = {k: torch.from_numpy(u[k]) for k in ('fc', 'att', 'att_masks', 'labels', 'masks')}
    return g, t


def _check_grads(P, g, key, rtol=3e-4, atol=3e-7):
    for k, p in P.items():
        if p.requires_grad:
            got = torch.zeros_like(p) if p.grad is None else p.grad
            np.testing.assert_allclose(got.numpy(), g['%s.grad.%s' % (key, k)], rtol=rtol, atol=atol, err_msg=k)


def _updown_drops(masks, B, K, N, steps, E, R, am, extra_steps=0):
    """Call order of AttModel in train mode: fc_embed's Dropout, att_embed's (packed when masked), then per step the embed
    Dropout (AttModel.py:74-76) and F.dropout(h_lang) (AttModel.py:637)."""
    fc = masks.take((B, R), 0.5)
    att = _unpack_regions(masks.take(None, 0.5), am) if am is not None else masks.take((B, K, R), 0.5)
    xt, out = [], []
    for _ in range(steps):
        xt.append(masks.take((N, E), 0.5))
        out.append(masks.take((N, R), 0.5))
    for _ in range(extra_steps):                     # the core step at t = L whose output _sample throws away (AttModel.py:288,335)
        masks.take((N, E), 0.5), masks.take((N, R), 0.5)
    return O.Drops(fc=fc, att=att, xt=torch.stack(xt), out=torch.stack(out))


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_train_mode_updown_dropout_sites_are_the_references(tag):
    g, t = _train_inputs()
    z, P = load('updown_tiny.npz')
    for v in P.values():
        v.requires_grad_(True)
    am = t['att_masks'] if tag == 'mask' else None
    key = 'updown.xe_' + tag
    want = g[key + '.logp']
    steps = int((np.abs(want).sum((0, 2)) > 0).sum())                  # columns the reference wrote before its all-pad break
    B, K = t['att'].shape[:2]
    N = want.shape[0]
    masks = _Masks(g, key)
    drops = _updown_drops(masks, B, K, N, steps, 16, 16, am)
    assert masks.left() == 0, '%d recorded masks not consumed: the reference drops a tensor the oracle does not' % masks.left()
    logp = O.forward_teacher(P, t['fc'], t['att'], t['labels'][..., :-1], am, drops)
    np.testing.assert_allclose(logp.detach().numpy(), want, rtol=1e-5, atol=3e-6)
    assert np.abs(want - z['xe_logp_' + tag]).max() > 1e-2             # dropout really changed the outputs
    loss = O.lm_criterion(logp, t['labels'][..., 1:], t['masks'][..., 1:])
    np.testing.assert_allclose(loss.item(), g[key + '.loss'], rtol=1e-5)
    loss.backward()
    _check_grads(P, g, key)


def test_train_mode_updown_sampled_rollout_dropout_sites():
    """AttModel._sample in train() (the SCST rollout of BASELINE configs[2] runs at drop_prob_lm 0.5): teacher-forcing the tokens
    the reference drew, under the masks it drew, reproduces its dense log-probs, RewardCriterion loss and every gradient."""
    g, t = _train_inputs()
    z, P = load('updown_tiny.npz')
    for v in P.values():
        v.requires_grad_(True)
    key = 'updown.sample'
    forced = torch.from_numpy(g[key + '.seq'])
    N, L = forced.shape
    B, K = t['att'].shape[:2]
    masks = _Masks(g, key)
    ran = (len(masks.m) - 2) // 2                                       # core steps the reference ran (L + 1 when no early exit)
    drops = _updown_drops(masks, B, K, N, min(ran, L), 16, 16, t['att_masks'], extra_steps=max(0, ran - L))
    assert masks.left() == 0 and ran in (L, L + 1) or bool((forced[:, -1] == 0).all())
    seq, slp = O.rollout(P, t['fc'], t['att'], t['att_masks'], method='sample', sample_n=2, max_len=L, drops=drops, forced=forced)
    assert np.array_equal(seq.numpy(), g[key + '.seq'])
    np.testing.assert_allclose(slp.detach().numpy(), g[key + '.logp'], rtol=1e-5, atol=3e-6)
    loss = O.reward_criterion(slp, seq, torch.from_numpy(g[key + '.reward']))
    np.testing.assert_allclose(loss.item(), g[key + '.loss'], rtol=1e-5)
    loss.backward()
    _check_grads(P, g, key)


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_train_mode_newfc_dropout_sites_are_the_references(tag):
    """LSTMCore.dropout (FCModel.py:23,40) on next_h only -- the state keeps the undropped h -- and also drawn (then discarded
    with the output) for the image-feeding call of step 0 (AttModel.py:925-927)."""
    g, t = _train_inputs()
    z, P = load('newfc_tiny.npz')
    for v in P.values():
        v.requires_grad_(True)
    key = 'newfc.xe_' + tag
    want = g[key + '.logp']
    steps = int((np.abs(want).sum((0, 2)) > 0).sum())
    N = want.shape[0]
    masks = _Masks(g, key)
    masks.take((N, 16), 0.5)                                            # the image feed's output is thrown away
    drop_out = torch.stack([masks.take((N, 16), 0.5) for _ in range(steps)])
    assert masks.left() == 0
    logp = O.newfc_forward_teacher(P, t['fc'], t['labels'][..., :-1], drop_out)
    np.testing.assert_allclose(logp.detach().numpy(), want, rtol=1e-5, atol=3e-6)
    assert np.abs(want - z['xe_logp']).max() > 1e-2
    loss = O.lm_criterion(logp, t['labels'][..., 1:], t['masks'][..., 1:])
    np.testing.assert_allclose(loss.item(), g[key + '.loss'], rtol=1e-5)
    loss.backward()
    _check_grads(P, g, key)


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_train_mode_transformer_dropout_sites_are_the_references(tag):
    """att_embed's Dropout(drop_prob_lm) on the packed regions, then make_model's one rate (0.1) on: attention probabilities
    (TransformerModel.py:152-162), every SublayerConnection output before the residual add (:89-101), the FFN hidden after
    its ReLU (:205-206) and embedding + positional encoding (:231-233) -- 22 calls at N_enc = N_dec = 2, in this order."""
    from oracle import transformer as T
    g, t = _train_inputs()
    z = np.load(os.path.join(GOLDEN, 'transformer_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith('.pe'):
            v.requires_grad_(True)
    am = t['att_masks'] if tag == 'mask' else None
    key = 'transformer.xe_' + tag
    masks = _Masks(g, key)
    drop = _hook(masks, am, (('att_embed', 0.5), ('', 0.1)))
    logp = T.forward_teacher(P, t['att'], t['labels'][..., :-1], am, h=2, n_enc=2, n_dec=2, drop=drop)
    assert masks.left() == 0, '%d recorded masks not consumed' % masks.left()
    np.testing.assert_allclose(logp.detach().numpy(), g[key + '.logp'], rtol=1e-5, atol=3e-6)
    assert np.abs(g[key + '.logp'] - z['xe_logp_' + tag]).max() > 1e-2
    loss = O.lm_criterion(logp, t['labels'][..., 1:], t['masks'][..., 1:])
    np.testing.assert_allclose(loss.item(), g[key + '.loss'], rtol=1e-5)
    loss.backward()
    _check_grads(P, g, key)


_AOA_RATES = (('att_embed', 0.5), ('xt', 0.5), ('ctx', 0.5), ('out', 0.5), ('.aoa', 0.3), ('.attn', 0.1), ('.res', 0.1))


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_train_mode_aoa_dropout_sites_are_the_references(tag):
    """att_embed (0.5, packed); per refiner layer: attention probabilities (0.1, AoAModel.py:18,53), dropout_aoa on cat[att, q]
    (0.3, :42-44,92), SublayerConnection (0.1, :119); per decode step: embed (0.5), ctx_drop on state[0][1] (:158-160,165),
    the decoder attention's probabilities (0.1), out_drop on the output but not on the stored state (:185)."""
    from oracle import aoa as A
    g, t = _train_inputs()
    z = np.load(os.path.join(GOLDEN, 'aoa_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]).requires_grad_(True) for k in z.files if k.startswith('P.')}
    am = t['att_masks'] if tag == 'mask' else None
    key = 'aoa.xe_' + tag
    masks = _Masks(g, key)
    logp = A.forward_teacher(P, t['att'], t['labels'][..., :-1], am, h=2, drop=_hook(masks, am, _AOA_RATES))
    assert masks.left() == 0, '%d recorded masks not consumed' % masks.left()
    np.testing.assert_allclose(logp.detach().numpy(), g[key + '.logp'], rtol=1e-5, atol=3e-6)
    assert np.abs(g[key + '.logp'] - z['xe_logp_' + tag]).max() > 1e-2
    loss = O.lm_criterion(logp, t['labels'][..., 1:], t['masks'][..., 1:])
    np.testing.assert_allclose(loss.item(), g[key + '.loss'], rtol=1e-5)
    loss.backward()
    _check_grads(P, g, key)


def test_train_mode_aoa_sampled_rollout_dropout_sites():
    """The new_self_critical rollout of BASELINE configs[4] (AttModel._sample in train()): forced to the tokens the reference drew
    under the masks it drew.  The reference runs one more core step at t = L and discards it: exactly its 4 masks stay unused."""
    from oracle import aoa as A
    g, t = _train_inputs()
    z = np.load(os.path.join(GOLDEN, 'aoa_tiny.npz'))
    P = {k[2:]: torch.from_numpy(z[k]).requires_grad_(True) for k in z.files if k.startswith('P.')}
    key = 'aoa.sample'
    forced = torch.from_numpy(g[key + '.seq'])
    masks = _Masks(g, key)
    seq, slp = A.sample(P, t['att'], t['att_masks'], 2, forced.shape[1], n=2, drop=_hook(masks, t['att_masks'], _AOA_RATES), forced=forced)
    ended_early = bool((forced[:, -1] == 0).all())
    assert masks.left() == (0 if ended_early else 4)
    assert np.array_equal(seq.numpy(), g[key + '.seq'])
    np.testing.assert_allclose(slp.detach().numpy(), g[key + '.logp'], rtol=1e-5, atol=3e-6)
    loss = O.reward_criterion(slp, seq, torch.from_numpy(g[key + '.reward']))
    np.testing.assert_allclose(loss.item(), g[key + '.loss'], rtol=1e-5)
    loss.backward()
    _check_grads(P, g, key)
