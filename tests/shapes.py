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
