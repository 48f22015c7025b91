"""graph_step.TrainStep: the training iteration of the two data-parallel BASELINE configurations (configs[3] Transformer XE,
configs[4] AoA new-self-critical; reference tools/train.py:185-196) captured into a hipGraph must be the SAME computation as the
stepped iteration: same loss to the last bit at every step, same parameters after 12 steps, same random streams (the epoch word of
capmi_step_state), same Adam bias corrections (capmi_adam_step_dyn)."""
import numpy as np
import pytest
import torch

import shapes

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _refs(rng, V1, L, n=4):
    """n reference captions [n, L] (0-padded) over a small vocabulary: common tokens repeat, so CIDEr-D scores are not all zero"""
    rows = np.zeros((n, L), dtype=np.int64)
    for r in range(n):
        ln = int(rng.integers(2, L))
        rows[r, :ln] = rng.integers(1, min(V1, 12), ln)
    return rows


def _setup(family, seed=77):
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper
    from imagecaptioning.pytorch_amd.captioning.utils import rewards
    V1, L, F = 61, 6, 40
    common = dict(seq_length=L, max_length=L, vocab_size=V1 - 1, fc_feat_size=F, att_feat_size=F,
                  vocab={str(i): 'w%d' % i for i in range(1, V1)}, learning_rate=1e-3, grad_clip_value=0.1)
    if family == 'transformer':
        opt = synthetic.updown_opt(caption_model='transformer', input_encoding_size=32, rnn_size=64, d_model=32, d_ff=64, N_enc=2,
                                   N_dec=2, num_att_heads=4, dropout=0.1, drop_prob_lm=0.5, **common)
    else:
        opt = synthetic.updown_opt(caption_model='aoa', input_encoding_size=32, rnn_size=32, att_hid_size=16, num_heads=4,
                                   multi_head_scale=1, use_multi_head=2, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
                                   mean_feats=1, ctx_drop=1, dropout_aoa=0.3, drop_prob_lm=0.5, train_sample_n=3,
                                   structure_loss_type='new_self_critical', structure_loss_weight=1.0, **common)
    torch.manual_seed(seed)
    model = models.setup(opt).to(DEV)
    model.train()
    flat = model.flatten_parameters_()
    lw = LossWrapper(model, opt)
    rewards.reset_scorer()
    if family == 'aoa':
        rng = np.random.default_rng(3)
        corpus = [_refs(rng, V1, L) for _ in range(50)]
        df, ref_len = synthetic.document_frequency(corpus)
        rewards.init_scorer((df, ref_len), device=torch.device(DEV))
    return opt, model, flat, lw, (V1, L, F)


def _batches(family, dims, nb=3, B=3, K=5):
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning.utils import rewards
    V1, L, F = dims
    out = []
    g = torch.Generator().manual_seed(5)
    for b in range(nb):
        fc, att = shapes.feats(B, K=K, F=F, seed=10 + b)
        d = {'fc_feats': fc.to(DEV), 'att_feats': att.to(DEV), 'att_masks': None, 'labels': None, 'masks': None, 'gts': None}
        if family == 'transformer':
            n, T = 2, L + 2
            labels = torch.zeros(B, n, T, dtype=torch.long)
            masks = torch.zeros(B, n, T)
            for i in range(B):
                for j in range(n):
                    ln = int(torch.randint(2, L, (1,), generator=g))
                    labels[i, j, 1:1 + ln] = torch.randint(1, V1, (ln,), generator=g)
                    masks[i, j, :ln + 2] = 1
            d['labels'], d['masks'] = labels.to(DEV), masks.to(DEV)
        else:
            rng = np.random.default_rng(40 + b)
            d['gts'] = rewards.pack_gts([_refs(rng, V1, L) for _ in range(B)])
        out.append(d)
    return out


@pytest.mark.parametrize('family,flags', [('transformer', (False, False)), ('aoa', (False, True))])
def test_captured_training_step_is_the_stepped_step_bit_for_bit(family, flags):
    from imagecaptioning.pytorch_amd.graph_step import TrainStep
    runs = {}
    for mode in ('stepped', 'graph'):
        opt, model, flat, lw, dims = _setup(family)
        batches = _batches(family, dims)
        ts = TrainStep(lw, flat, opt, DEV, graph=(mode == 'graph'))
        losses = []
        for it in range(12):
            lr = 1e-3 if it < 6 else 5e-4                     # the schedule moves the rate: capmi_step_set_lr, outside the graph
            loss, out = ts(batches[it % len(batches)], flags[0], flags[1], lr=lr)
            losses.append(loss.clone())
        torch.cuda.synchronize()
        st = ts.state.read()
        assert st.adam_step == 12 and st.epoch == 12 and flat.step_count == 12
        if mode == 'graph':
            assert ts.failed is None, ts.failed
            assert ts.captures == 1 and ts.replays == 11 and ts.stepped == 1, (ts.captures, ts.replays, ts.stepped)
        else:
            assert ts.captures == 0 and ts.stepped == 12
        runs[mode] = (torch.stack(losses).cpu(), flat.flat.clone().cpu(), flat.exp_avg_sq.clone().cpu())
    a, b = runs['stepped'], runs['graph']
    assert torch.isfinite(a[0]).all() and float(a[0].abs().sum()) > 0
    assert len(set(a[0].tolist())) > 6                      # the steps really differ (batches rotate, dropout moves, weights move)
    assert torch.equal(a[0], b[0]), (a[0], b[0])            # identical loss at every step, to the last bit
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])      # (r6: no atomics left in either step, csrc/embed_bwd_det.h)


def test_epoch_word_moves_the_dropout_stream_and_unbound_launches_are_unchanged():
    """capmi_rng_bind_epoch: the same seed argument draws a different mask at every epoch, the same mask at the same epoch, and the
    seed alone (the pre-r6 behaviour every injected-mask test relies on) when nothing is bound."""
    from imagecaptioning.pytorch_amd import ops
    dev = torch.device(DEV)
    base = ops.dropout_mask((4096,), 0.5, 1234, 0, dev).clone()
    st = ops.StepState(dev)
    with st.bound():
        m0 = ops.dropout_mask((4096,), 0.5, 1234, 0, dev).clone()          # epoch 0: seed + 0
        st.advance(0.9, 0.999)
        m1 = ops.dropout_mask((4096,), 0.5, 1234, 0, dev).clone()
        m1b = ops.dropout_mask((4096,), 0.5, 1234, 0, dev).clone()
        st.advance(0.9, 0.999)
        m2 = ops.dropout_mask((4096,), 0.5, 1234, 0, dev).clone()
    after = ops.dropout_mask((4096,), 0.5, 1234, 0, dev)
    assert torch.equal(base, m0) and torch.equal(base, after)
    assert torch.equal(m1, m1b)
    assert not torch.equal(m0, m1) and not torch.equal(m1, m2)
    s = st.read()
    assert s.epoch == 2 and s.adam_step == 2
    b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))           # the C ABI takes the betas as float
    assert abs(s.bc1 - (1 - b1 ** 2)) < 1e-7 and abs(s.bc2_sqrt - (1 - b2 ** 2) ** 0.5) < 1e-7


def test_adam_step_dyn_is_adam_step():
    from imagecaptioning.pytorch_amd import ops
    dev = torch.device(DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    n = 64 * 1000
    p0, gr = torch.randn(n, device=dev, generator=g), torch.randn(n, device=dev, generator=g) * 0.3
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    st = ops.StepState(dev, lr=0.0)
    for step in range(1, 5):
        lr = 1e-3 / step
        ops.adam_step(pa, gr, ma, va, lr, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.5, step)
        st.set_lr(lr)
        st.advance(0.9, 0.999)
        ops.adam_step_dyn(pb, gr, mb, vb, st, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.5)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
