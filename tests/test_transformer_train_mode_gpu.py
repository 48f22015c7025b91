"""Transformer XE in TRAIN mode against oracle/transformer.py under one injected dropout realisation (BASELINE configs[3] trains at
dropout 0.1 / drop_prob_lm 0.5 with 5 captions per image).

The reference's `_forward` repeats the embedded regions BEFORE the encoder (TransformerModel.py:316-321, 343-345): each of the n caption
rows of an image runs the encoder under its OWN dropout masks.  `tests/golden/train_mode.npz` pins that to the reference itself (its
recorded attention mask of encoder layer 0 is [B * n, h, K, K]); `oracle/transformer.forward_teacher(encode_per_caption=True)` restates it
and is checked against that fixture on CPU (tests/test_oracle_golden.py).  Here: the HIP engine == that oracle, with the masks the
engine draws (Philox) handed to the oracle's hooks."""
import numpy as np
import pytest
import torch

import shapes

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _engine_masks(E, model, seed, B, N, K, T, d, dff, h, nl, per_caption):
    """the realisation a TransformerGraph of `seed` draws, in its order (encode, then decoder_masks), named like the oracle's hooks"""
    dev = torch.device(DEV)
    de = E.Dropper(model.drop_prob_lm, seed, dev, True)
    dd = E.Dropper(model.dropout, seed ^ 0x5bd1e995, dev, True)
    Be = N if per_caption else B
    named = {'att_embed': de(B * K, d).view(B, K, d)}
    for i in range(nl):
        named['enc%d.attn' % i] = dd(Be, h, K, K)
        named['enc%d.res0' % i] = dd(Be * K, d).view(Be, K, d)
        named['enc%d.ff' % i] = dd(Be * K, dff).view(Be, K, dff)
        named['enc%d.res1' % i] = dd(Be * K, d).view(Be, K, d)
    named['tgt_embed'] = dd(N, T, d)
    for i in range(nl):
        named['dec%d.self.attn' % i] = dd(N, h, T, T)
        named['dec%d.res0' % i] = dd(N * T, d).view(N, T, d)
        named['dec%d.src.attn' % i] = dd(N, h, T, K)
        named['dec%d.res1' % i] = dd(N * T, d).view(N, T, d)
        named['dec%d.ff' % i] = dd(N * T, dff).view(N, T, dff)
        named['dec%d.res2' % i] = dd(N * T, d).view(N, T, d)
    return {k: v.cpu() for k, v in named.items()}


@pytest.mark.parametrize('tied', [False, True])
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('d,h,dff,nl,K,V1,B,n,L,F', [(48, 4, 100, 2, 7, 101, 3, 2, 6, 44),
                                                   (512, 8, 2048, 6, 36, 9488, 2, 3, 5, 2048)])      # configs/transformer sizes
def test_transformer_xe_train_mode_encoder_runs_per_caption_row(d, h, dff, nl, K, V1, B, n, L, F, masked, tied):
    from oracle import transformer as T, att_lstm as O
    from imagecaptioning.pytorch_amd import synthetic, transformer_engine as E
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion
    opt = synthetic.updown_opt(caption_model='transformer', input_encoding_size=d, rnn_size=dff, d_model=d, d_ff=dff, N_enc=nl,
                               N_dec=nl, num_att_heads=h, dropout=0.2, drop_prob_lm=0.3, seq_length=L, max_length=L,
                               vocab_size=V1 - 1, fc_feat_size=F, att_feat_size=F,
                               vocab={str(i): 'w%d' % i for i in range(1, V1)})
    torch.manual_seed(4712)
    model = models.setup(opt).to(DEV)
    model.train()
    model.tie_encoder_dropout = tied            # opt-in: ONE encoder pass per image (masks shared by its n caption rows)
    _, att = shapes.feats(B, K=K, F=F, seed=6)
    am = None
    if masked:
        am = torch.ones(B, K)
        am[0, K // 2:] = 0
    g = torch.Generator().manual_seed(31)
    N, Tt = B * n, L + 1
    labels = torch.zeros(B, n, Tt + 1, dtype=torch.long)
    for b in range(B):
        for j in range(n):
            ln = int(torch.randint(2, L, (1,), generator=g))
            labels[b, j, 1:1 + ln] = torch.randint(1, V1, (ln,), generator=g)
    lmask = torch.zeros(B, n, Tt + 1)
    for b in range(B):
        for j in range(n):
            lmask[b, j, :int((labels[b, j] > 0).sum()) + 2] = 1

    model._rng_calls = 0
    logp = model(None, att.to(DEV), labels[..., :-1].to(DEV), None if am is None else am.to(DEV))
    loss = LanguageModelCriterion()(logp, labels[..., 1:].to(DEV), lmask[..., 1:].to(DEV))
    model.zero_grad()
    loss.backward()

    model._rng_calls = 0
    named = _engine_masks(E, model, model._next_seed(), B, N, K, Tt, d, dff, h, nl, per_caption=not tied)
    used = set()

    def drop(name, x):
        used.add(name)
        assert tuple(named[name].shape) == tuple(x.shape), (name, tuple(named[name].shape), tuple(x.shape))
        return x * named[name]

    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for v in P.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    P['model.tgt_embed.1.pe'] = model.model.tgt_embed[1].pe.cpu()
    T.RELU_TIES = ties = {}
    try:
        want = T.forward_teacher(P, att, labels[..., :-1], am, h=h, n_enc=nl, n_dec=nl, drop=drop, encode_per_caption=not tied)
    finally:
        T.RELU_TIES = None
    assert used == set(named)
    assert float((logp.detach().cpu() - want.detach()).abs().max()) <= 2e-4
    want_loss = O.lm_criterion(want, labels[..., 1:], lmask[..., 1:])
    assert abs(float(loss.detach()) - float(want_loss.detach())) <= 1e-4 * max(1.0, abs(float(want_loss.detach())))
    want_loss.backward()
    for k, prm in model.named_parameters():
        w = P[k].grad
        tie = ties.get(k.rsplit('.w_1.', 1)[0]) if (k.endswith('.w_1.weight') or k.endswith('.w_1.bias')) else None
        err = (prm.grad.cpu() - w).abs()
        if tie is not None and bool(tie.any()):         # a unit on the ReLU's kink takes one token's whole term either way
            err_rows = err.reshape(err.shape[0], -1).max(1)[0]
            assert float(err_rows[tie].max()) <= 5e-2 * float(w.abs().max()) + 1e-6, (k, 'rows with a ReLU tie')
            err = err_rows[~tie]
        assert float(err.max()) <= 1e-3 * float(w.abs().max()) + 1e-6, k


def test_transformer_eval_and_dropout_free_training_encode_once_per_image():
    """With dropout off the n encoder rows of an image are identical copies: the engine encodes per image (B rows) -- the graph's
    encoder batch says so -- and the numbers are the reference's (the dropout-0 fixtures of test_full_size_parity_gpu.py)."""
    from imagecaptioning.pytorch_amd import synthetic, transformer_engine as E
    from imagecaptioning.pytorch_amd.captioning import models
    seen = []
    orig = E.TransformerGraph.encode

    def spy(self, att_feats, att_masks, rows_per_image=1):
        seen.append(rows_per_image)
        return orig(self, att_feats, att_masks, rows_per_image)

    E.TransformerGraph.encode = spy
    try:
        for drop, training, want in ((0.0, True, 1), (0.1, False, 1), (0.1, True, 2)):
            opt = synthetic.updown_opt(caption_model='transformer', input_encoding_size=32, rnn_size=64, d_model=32, d_ff=64, N_enc=1,
                                       N_dec=1, num_att_heads=2, dropout=drop, drop_prob_lm=0.5, seq_length=4, max_length=4,
                                       vocab_size=20, fc_feat_size=24, att_feat_size=24, vocab={str(i): 'w%d' % i for i in range(1, 21)})
            model = models.setup(opt).to(DEV).train(training)
            _, att = shapes.feats(2, K=5, F=24, seed=1)
            seq = torch.randint(1, 20, (2, 2, 5))
            seq[..., 0] = 0
            del seen[:]
            with torch.set_grad_enabled(training):
                model(None, att.to(DEV), seq.to(DEV), None)
            assert seen == [want], (drop, training, seen)
    finally:
        E.TransformerGraph.encode = orig
