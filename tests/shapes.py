"""Deterministic BASELINE-size inputs shared by the fixture generator (tests/golden/make_golden.py) and the GPU parity
tests: everything is drawn from seeded torch CPU generators, so the build container and the GPU box see the same bits."""
import torch


def full_size_params(seed=1234, V1=9488, R=1000, E=1000, A=512, F=2048):
    """UpDown parameters at configs/updown/updown.yml sizes under the reference's state_dict keys."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s, a: (torch.rand(*s, generator=g) * 2 - 1) * a       # noqa: E731
    P = {'embed.0.weight': torch.randn(V1, E, generator=g),
         'fc_embed.0.weight': u(R, F, a=F ** -0.5), 'fc_embed.0.bias': u(R, a=F ** -0.5),
         'att_embed.0.weight': u(R, F, a=F ** -0.5), 'att_embed.0.bias': u(R, a=F ** -0.5),
         'ctx2att.weight': u(A, R, a=R ** -0.5), 'ctx2att.bias': u(A, a=R ** -0.5),
         'core.att_lstm.weight_ih': u(4 * R, 2 * R + E, a=R ** -0.5), 'core.att_lstm.weight_hh': u(4 * R, R, a=R ** -0.5),
         'core.att_lstm.bias_ih': u(4 * R, a=R ** -0.5), 'core.att_lstm.bias_hh': u(4 * R, a=R ** -0.5),
         'core.lang_lstm.weight_ih': u(4 * R, 2 * R, a=R ** -0.5), 'core.lang_lstm.weight_hh': u(4 * R, R, a=R ** -0.5),
         'core.lang_lstm.bias_ih': u(4 * R, a=R ** -0.5), 'core.lang_lstm.bias_hh': u(4 * R, a=R ** -0.5),
         'core.attention.h2att.weight': u(A, R, a=R ** -0.5), 'core.attention.h2att.bias': u(A, a=R ** -0.5),
         'core.attention.alpha_net.weight': u(1, A, a=A ** -0.5), 'core.attention.alpha_net.bias': u(1, a=A ** -0.5),
         'logit.weight': u(V1, R, a=R ** -0.5), 'logit.bias': u(V1, a=R ** -0.5)}
    return P


def feats(B, K=36, F=2048, seed=1):
    g = torch.Generator().manual_seed(seed)
    fc = (torch.randn(B, F, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, K, F, generator=g) * 0.5).clamp_min(0)
    return fc, att


def c3_case(seed=2):
    """BASELINE configs[2] shape: bs10 x train_sample_n 5, L=20, dropout 0.5 masks + Gumbel noise + a reward, all seeded."""
    from oracle import att_lstm as O
    B, n, K, L, R, E, V1 = 10, 5, 36, 20, 1000, 1000, 9488
    N = B * n
    g = torch.Generator().manual_seed(seed)
    fc = (torch.randn(B, 2048, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, K, 2048, generator=g) * 0.5).clamp_min(0)
    drops = O.make_drops(0.5, B, K, N, L, E, R, g)
    gum = -torch.log(-torch.log(torch.rand(L, N, V1, generator=g).clamp_min(1e-20)))
    reward = torch.randn(N, 1, generator=g).repeat(1, L)
    return dict(B=B, n=n, K=K, L=L, N=N, fc=fc, att=att, drops=drops, gumbel=gum, reward=reward)


def c2_labels(B=10, n=5, L=20, V1=9488, seed=5):
    """labels [B,n,L+2] / masks as dataloader.py:245-249 builds them (BOS/EOS columns 0, nonzeros + 2 ones)."""
    g = torch.Generator().manual_seed(seed)
    labels = torch.zeros(B, n, L + 2, dtype=torch.long)
    masks = torch.zeros(B, n, L + 2)
    for b in range(B):
        for j in range(n):
            ln = L if (b == 0 and j == 0) else int(torch.randint(6, L + 1, (1,), generator=g))
            labels[b, j, 1:ln + 1] = torch.randint(1, V1, (ln,), generator=g)
            masks[b, j, :ln + 2] = 1
    return labels, masks


def grad_probe(t, k=256):
    """a fixed, spread-out sample of a tensor's elements (what the compact fixtures store of each gradient)."""
    flat = t.reshape(-1)
    step = max(1, flat.numel() // k)
    return flat[::step][:k]


def seeded_state(shapes, seed):
    """Deterministic parameters for ANY model from its state_dict shapes ({name: shape}): the fixture generator (reference
    model, build container) and the GPU test (HIP mirror, same keys and shapes) draw identical weights without a 200 MB file.
    Matrices uniform +-fan_in^-0.5, LayerNorm gains around 1, other vectors small."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if len(shp) >= 2:
            a = float(shp[-1]) ** -0.5
            out[name] = (torch.rand(*shp, generator=g) * 2 - 1) * a
        elif name.endswith('a_2') or name.endswith('norm.weight') or name.endswith('layer_norm.weight'):
            out[name] = 1 + 0.1 * torch.randn(*shp, generator=g)
        else:
            out[name] = 0.02 * torch.randn(*shp, generator=g)
    return out


def big_opt(family):
    """configs/transformer/transformer.yml and configs/aoa.yml sizes on the synthetic vocabulary (dropouts off: the reference's
    RNG stream cannot be reproduced by a kernel)."""
    from imagecaptioning.pytorch_amd import synthetic
    if family == 'transformer':
        return synthetic.updown_opt(caption_model='transformer', input_encoding_size=512, rnn_size=2048, d_model=512, d_ff=2048,
                                    N_enc=6, N_dec=6, num_att_heads=8, dropout=0.0, drop_prob_lm=0.0)
    return synthetic.updown_opt(caption_model='aoa', input_encoding_size=1024, rnn_size=1024, att_hid_size=512, num_heads=8,
                                multi_head_scale=1, use_multi_head=2, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
                                mean_feats=1, ctx_drop=1, dropout_aoa=0.3, drop_prob_lm=0.0)


def ragged_masks(B, K=36, seed=1, lo=10):
    """att_masks as the loader builds them for images with fewer regions than the batch maximum (dataloader.py:221-229):
    ones up to the image's region count, image 0 keeps all K (the batch maximum)."""
    g = torch.Generator().manual_seed(seed + 1000)
    am = torch.zeros(B, K)
    for b in range(B):
        k = K if b == 0 else int(torch.randint(lo, K + 1, (1,), generator=g))
        am[b, :k] = 1
    return am


# The config-size beam fixtures (make_golden.py beam5).  Random-init logits are nearly uniform over 9 488 words: the b-th and
# (b+1)-th of 47 440 candidates are then 1e-5 apart (a coin flip between two fp32 implementations) and no beam ever ends.  The
# logit layer is therefore scaled to a trained model's sharpness (about -2.3 nats per chosen word) and the EOS bias raised so
# that EOS enters the top 5 at the first steps for some images / beams and never for others ('long': no EOS at all).
BEAM5_LOGIT_SCALE = 100.0
BEAM5_EOS_BIAS = {('aoa', 'end'): 16.0, ('updown', 'end'): 4.5, ('aoa', 'long'): 0.0, ('updown', 'long'): 0.0}
# seeds: picked among 31..36 (aoa) and 41..46 (updown) so that every live top-5 / top-6 gap the reference saw is >= 1e-4 (>= 1e-3 but for two aoa cases) (make_golden.py
# beam5 prints and stores the smallest gap; fp32 sums near -40 carry about 1e-5 of rounding)
BEAM5_SEED = {'aoa': 34, 'updown': 44}


def beam5_state(name, state_shapes, seed, eos='end'):
    """weights of the beam-5 fixtures: updown -> full_size_params, aoa -> seeded_state; logit layer sharpened, EOS bias raised."""
    st = full_size_params(seed=seed) if name == 'updown' else seeded_state(state_shapes, seed)
    st['logit.weight'] = st['logit.weight'] * BEAM5_LOGIT_SCALE
    st['logit.bias'] = st['logit.bias'].clone()
    st['logit.bias'][0] += BEAM5_EOS_BIAS[(name, eos)]
    return st


# ---- beams that END IN THE MIDDLE of the sequence at config size (`make_golden.py beam5mid`).  With a constant
# EOS bias a random-init decoder ends its beams at steps 0-3 or never (its state reaches a fixed point, so EOS is either in the
# top 5 from the start or not at all).  A trained model ends a caption because its state counts: the fixture gives every family a
# CLOCK feature -- one hidden unit (LSTM families: a cell that adds `rate` per step, h = tanh(rate * (t + 1))) or one residual
# channel (Transformer: a positional-encoding channel of wavelength ~100 that nothing else writes) -- and ties the EOS logit to
# that unit alone (weight w00, bias b0): EOS climbs into the top 5 somewhere inside the sequence, a beam ends, the others go on
# (CaptionModel.py:176-198: the -1000 bookkeeping with live and ended beams mixed over 47 440 candidates).
BEAM5_MID = {'updown': (0.3, -10.0, 0.1), 'aoa': (0.45, -20.0, 0.1), 'transformer': (1.0, -20.0, 0.0)}
BEAM5_MID_SEED = {'updown': 47, 'aoa': 40, 'transformer': 57}


def _lstm_clock(st, pre, R, rate, unit=0):
    """unit `unit` of the LSTMCell `pre` becomes c_t = c_{t-1} + ~rate (i = f = o ~ 1, g = tanh^-1-ish rate), h = tanh(c_t)"""
    import math
    for w in ('weight_ih', 'weight_hh'):
        W = st[pre + '.' + w] = st[pre + '.' + w].clone()
        for gate in range(4):
            W[gate * R + unit] *= 0.02 if gate == 2 else 0.0            # a little state dependence stays in the increment
    for gate, val in ((0, 10.0), (1, 10.0), (2, math.atanh(rate)), (3, 10.0)):
        for b in ('bias_ih', 'bias_hh'):
            v = st[pre + '.' + b] = st[pre + '.' + b].clone()
            v[gate * R + unit] = val / 2


def mid_state(name, state_shapes, seed, w00=None, b0=None, rate=None):
    """weights of the mid-ending beam fixtures (see above); sharpened logit layer as in beam5_state"""
    d = BEAM5_MID[name]
    w00, b0, rate = (d[0] if w00 is None else w00), (d[1] if b0 is None else b0), (d[2] if rate is None else rate)
    st = full_size_params(seed=seed) if name == 'updown' else seeded_state(state_shapes, seed)
    lw = 'model.generator.proj.weight' if name == 'transformer' else 'logit.weight'
    lb = lw[:-6] + 'bias'
    st[lw] = st[lw] * BEAM5_LOGIT_SCALE
    st[lb] = st[lb].clone()
    if name == 'updown':
        R = st['core.lang_lstm.weight_hh'].shape[1]
        _lstm_clock(st, 'core.lang_lstm', R, rate)
        clock = 0
    elif name == 'aoa':
        R = st['core.att_lstm.weight_hh'].shape[1]
        _lstm_clock(st, 'core.att_lstm', R, rate)
        # out[0] = GLU row 0 = (h_att[0]) * sigmoid(10): the clock passes the att2ctx layer untouched (AoAModel.py:176-181)
        W, b = st['core.att2ctx.0.weight'].clone(), st['core.att2ctx.0.bias'].clone()
        W[0] = 0.0
        W[0, R] = 1.0
        W[R] = 0.0
        b[0], b[R] = 0.0, 10.0
        st['core.att2ctx.0.weight'], st['core.att2ctx.0.bias'] = W, b
        clock = 0
    else:
        # residual channel 154 carries sin(t / 10000^(154/512)) = sin(0.0627 t) from the positional encoding (TransformerModel.py:
        # 224-240) and nothing else: the word embedding and every decoder sublayer's output projection leave it alone
        clock = 154
        import math
        d = st['model.tgt_embed.1.pe'].shape[-1]           # seeded_state drew noise for the buffer: put the real table back
        pos = torch.arange(0, st['model.tgt_embed.1.pe'].shape[-2]).unsqueeze(1).float()
        div = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
        pe = torch.zeros(pos.shape[0], d)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        st['model.tgt_embed.1.pe'] = pe.reshape(st['model.tgt_embed.1.pe'].shape)
        st['model.tgt_embed.0.lut.weight'] = st['model.tgt_embed.0.lut.weight'].clone()
        st['model.tgt_embed.0.lut.weight'][:, clock] = 0.0
        for k in list(st):
            if k.startswith('model.decoder.layers.') and (k.endswith('linears.3.weight') or k.endswith('w_2.weight')):
                st[k] = st[k].clone()
                st[k][clock] = 0.0
            elif k.startswith('model.decoder.layers.') and (k.endswith('linears.3.bias') or k.endswith('w_2.bias')):
                st[k] = st[k].clone()
                st[k][clock] = 0.0
        st['model.decoder.norm.a_2'] = st['model.decoder.norm.a_2'].clone()
        st['model.decoder.norm.a_2'][clock] = 10.0
        st['model.decoder.norm.b_2'] = st['model.decoder.norm.b_2'].clone()
        st['model.decoder.norm.b_2'][clock] = 0.0
    st[lw][0] = 0.0
    st[lw][0, clock] = w00 * BEAM5_LOGIT_SCALE
    st[lb][0] = b0
    return st
