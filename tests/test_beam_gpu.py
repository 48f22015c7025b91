"""Device beam search vs the fixture produced by the real reference (tests/golden/updown_tiny_beam.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from test_model_api_gpu import golden_model, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag,bs,masked,kw', [('b3', 3, False, {}), ('b2m', 2, True, {}), ('b3n', 3, False, {'sample_n': 3}),
                                              ('b3lp', 3, True, {'length_penalty': 'avg_0'})])
def test_beam_search_matches_reference(tag, bs, masked, kw):
    z, model = golden_model(False)
    g = np.load(os.path.join(GOLDEN, 'updown_tiny_beam.npz'))
    model.eval()
    fc, att = torch.from_numpy(z['fc']).to(DEV), torch.from_numpy(z['att']).to(DEV)
    am = torch.from_numpy(z['att_masks']).to(DEV) if masked else None
    o = {'sample_method': 'beam_search', 'beam_size': bs, 'sample_n': 1}
    o.update(kw)
    with torch.no_grad():
        seq, slp = model(fc, att, am, opt=o, mode='sample')
    assert np.array_equal(seq.cpu().numpy(), g[tag + '_seq'])
    np.testing.assert_allclose(slp.cpu().numpy(), g[tag + '_logp'], rtol=2e-5, atol=1e-5)
    for k, beams in enumerate(model.done_beams):
        assert len(beams) == int(g['%s_n%d' % (tag, k)])
        for j, bm in enumerate(beams):
            assert np.array_equal(bm['seq'].cpu().numpy(), g['%s_%d_%d_seq' % (tag, k, j)]), (k, j)
            np.testing.assert_allclose(bm['p'], g['%s_%d_%d_p' % (tag, k, j)], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(bm['unaug_p'], g['%s_%d_%d_unaug' % (tag, k, j)], rtol=1e-4)
            assert bm['logps'].shape == (bm['seq'].shape[0], slp.shape[2])


def test_beam_select_properties():
    """The invariants the reference asserts every step (CaptionModel.py:89,97,100): scores are the sorted top-b of
    sum + logp, parents/tokens decode the flat index, ended beams are pushed down by 1000."""
    from imagecaptioning.pytorch_amd import _lib
    from imagecaptioning.pytorch_amd._lib import lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(0)
    B, cur, bd, V1 = 4, 5, 5, 9488
    logp = torch.log_softmax(torch.randn(B * cur, V1, generator=g) * 2, 1).to(DEV)
    sums = (torch.randn(B, bd, generator=g) * 3).to(DEV)
    parent = torch.empty(B, bd, dtype=torch.int32, device=DEV)
    token = torch.empty(B, bd, dtype=torch.long, device=DEV)
    score = torch.empty(B, bd, device=DEV)
    nxt = torch.empty(B, bd, device=DEV)
    ended = torch.empty(B, bd, dtype=torch.uint8, device=DEV)
    logp[3, 0] = 5.0        # force an EOS candidate to win for image 0 (row 3 = image 0, beam 3)
    _lib.check(lib.capmi_beam_select(ptr(logp), ptr(sums), B, cur, bd, V1, 0, ptr(parent), ptr(token), ptr(score), ptr(nxt),
                                     ptr(ended), stream_ptr()), 'select')
    cand = (sums[:, :cur].unsqueeze(-1) + logp.view(B, cur, V1)).reshape(B, -1)
    ys, ix = torch.sort(cand, -1, True)
    assert torch.equal(score, ys[:, :bd])
    assert torch.equal(parent.long(), ix[:, :bd] // V1) and torch.equal(token, ix[:, :bd] % V1)
    assert int(token[0, 0]) == 0 and int(ended[0, 0]) == 1
    assert torch.allclose(nxt, score - 1000.0 * ended.float())


def _family_model(name):
    from imagecaptioning.pytorch_amd.captioning import models
    from test_model_api_gpu import tiny_opt
    z = np.load(os.path.join(GOLDEN, name + '_tiny.npz'))
    if name == 'transformer':
        opt = tiny_opt(caption_model='transformer', N_enc=2, N_dec=2, d_model=16, d_ff=32, num_att_heads=2, dropout=0.0)
    else:
        opt = tiny_opt(caption_model='aoa', refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA', use_multi_head=2, num_heads=2,
                       multi_head_scale=1, mean_feats=1, ctx_drop=1, dropout_aoa=0.3, num_layers=2)
    model = models.setup(opt)
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('P.')})
    return model.to(DEV).eval()


@pytest.mark.parametrize('name', ['transformer', 'aoa'])
@pytest.mark.parametrize('tag,bs,masked,kw', [('b3', 3, False, {}), ('b2m', 2, True, {}), ('b3n', 3, False, {'sample_n': 3}),
                                              ('b3lp', 3, True, {'length_penalty': 'avg_0'})])
def test_beam_search_transformer_aoa_match_reference(name, tag, bs, masked, kw):
    """BASELINE configs[4] evaluates AoA with beam_size 5: the host-stepped beam search (native selection / reorder /
    normalisation kernels, KV caches or LSTM state following the beams by parent pointer) against fixtures produced by
    the real reference's CaptionModel.beam_search on the same weights."""
    model = _family_model(name)
    u = np.load(os.path.join(GOLDEN, 'updown_tiny.npz'))
    g = np.load(os.path.join(GOLDEN, name + '_tiny_beam.npz'))
    att = torch.from_numpy(u['att']).to(DEV)
    am = torch.from_numpy(u['att_masks']).to(DEV) if masked else None
    o = {'sample_method': 'beam_search', 'beam_size': bs, 'sample_n': 1}
    o.update(kw)
    with torch.no_grad():
        seq, slp = model(None, att, am, opt=o, mode='sample')
    assert np.array_equal(seq.cpu().numpy(), g[tag + '_seq'])
    np.testing.assert_allclose(slp.cpu().numpy(), g[tag + '_logp'], rtol=5e-5, atol=2e-5)
    for k, beams in enumerate(model.done_beams):
        assert len(beams) == int(g['%s_n%d' % (tag, k)])
        for j, bm in enumerate(beams):
            assert np.array_equal(bm['seq'].cpu().numpy(), g['%s_%d_%d_seq' % (tag, k, j)]), (k, j)
            np.testing.assert_allclose(bm['p'], g['%s_%d_%d_p' % (tag, k, j)], rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(bm['unaug_p'], g['%s_%d_%d_unaug' % (tag, k, j)], rtol=1e-4)


@pytest.mark.parametrize('seed', range(8))
def test_beam_select_invariants_random_shapes(seed):
    """The reference's per-step asserts (CaptionModel.py:89,97,100) as properties over random shapes: vocabulary sizes that
    are not multiples of 4 or of the workgroup size, first step (cur = 1) and later steps (cur = bd), ties, -inf entries
    (decoding constraints) and beams already pushed down by -1000."""
    from imagecaptioning.pytorch_amd import _lib
    from imagecaptioning.pytorch_amd._lib import lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(100 + seed)
    B = int(torch.randint(1, 7, (1,), generator=g))
    bd = int(torch.randint(1, 9, (1,), generator=g))
    cur = 1 if seed % 3 == 0 else bd
    V1 = int(torch.randint(max(bd, 5), 3000, (1,), generator=g)) if seed % 2 else 9488
    logp = torch.log_softmax(torch.randn(B * cur, V1, generator=g) * 3, 1)
    logp[torch.rand(B * cur, V1, generator=g) < 0.01] = float('-inf')          # constrained entries
    if seed % 4 == 1:
        logp = (logp * 4).round() / 4                                           # exact ties: lowest flat index wins, like torch.sort(stable)
    sums = torch.randn(B, bd, generator=g) * 2
    sums[torch.rand(B, bd, generator=g) < 0.3] -= 1000.0                        # beams that already ended
    last = int(seed % 5 == 4)
    d = lambda t: t.to(DEV).contiguous()                                        # noqa: E731
    logp_d, sums_d = d(logp), d(sums)
    parent = torch.empty(B, bd, dtype=torch.int32, device=DEV)
    token = torch.empty(B, bd, dtype=torch.long, device=DEV)
    score, nxt = torch.empty(B, bd, device=DEV), torch.empty(B, bd, device=DEV)
    ended = torch.empty(B, bd, dtype=torch.uint8, device=DEV)
    _lib.check(lib.capmi_beam_select(ptr(logp_d), ptr(sums_d), B, cur, bd, V1, last, ptr(parent), ptr(token), ptr(score), ptr(nxt),
                                     ptr(ended), stream_ptr()), 'select')
    cand = (sums[:, :cur].unsqueeze(-1) + logp.view(B, cur, V1)).reshape(B, -1)
    ys, ix = torch.sort(cand, dim=-1, descending=True, stable=True)
    ys, ix = ys[:, :bd], ix[:, :bd]
    assert torch.equal(score.cpu(), ys)                                         # CaptionModel.py:100: beam_logprobs_sum == ys
    finite = torch.isfinite(ys)
    got_ix = parent.cpu().long() * V1 + token.cpu()
    if seed % 4 != 1:                                                           # without ties the choice itself is unique
        assert torch.equal(got_ix[finite], ix[finite])
    else:                                                                       # with ties: the chosen candidates carry the sorted scores
        assert torch.equal(cand.gather(1, got_ix)[finite], ys[finite])
        assert all(len(set(r.tolist())) == bd for r in got_ix)                  # and no candidate is taken twice
    tok = token.cpu()
    want_end = (tok == 0) | bool(last)
    assert torch.equal(ended.cpu().bool(), want_end)
    assert torch.equal(nxt.cpu()[finite], (ys - 1000.0 * want_end.float())[finite])


@pytest.mark.parametrize('name', ['aoa', 'updown'])
@pytest.mark.parametrize('tag,masked,kw,eos', [('b5', False, {}, 'end'), ('b5m', True, {}, 'end'), ('b5n', False, {'sample_n': 5}, 'end'),
                                               ('b5long', True, {}, 'long')])
def test_beam5_at_config_size_vs_the_reference_itself(name, tag, masked, kw, eos):
    """BASELINE configs[4] evaluates with beam_size 5 (MODEL_ZOO.md:3; VERDICT r3 missing #1(ii)): the segmented top-5 over
    5 x 9 488 = 47 440 candidates per image and step, L = 20, B = 3 images, at configs/aoa.yml and configs/updown sizes, against
    the outputs of the reference's own AttModel._sample_beam / CaptionModel.beam_search (AttModel.py:218-256,
    CaptionModel.py:35-209) on the same seeded weights (tests/golden/beam5_config_size.npz, `make_golden.py beam5`): seq and
    every done beam's tokens exact, log-probs / p / unaug_p within fp32 tolerance."""
    import shapes
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    g = np.load(os.path.join(GOLDEN, 'beam5_config_size.npz'))
    opt = shapes.big_opt('aoa') if name == 'aoa' else synthetic.updown_opt(drop_prob_lm=0.0)
    model = models.setup(opt)
    seed = shapes.BEAM5_SEED[name]
    model.load_state_dict(shapes.beam5_state(name, {k: v.shape for k, v in model.state_dict().items()}, seed, eos))
    model = model.to(DEV).eval()
    B, bs = 3, 5
    fc, att = shapes.feats(B, seed=seed)
    am = shapes.ragged_masks(B, seed=seed).to(DEV) if masked else None
    o = {'sample_method': 'beam_search', 'beam_size': bs, 'sample_n': 1}
    o.update(kw)
    with torch.no_grad():
        seq, slp = model(fc.to(DEV), att.to(DEV), am, opt=o, mode='sample')
    t = name + '_' + tag
    assert np.array_equal(seq.cpu().numpy(), g[t + '_seq'])
    sel = slp.gather(2, seq.unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(sel, g[t + '_sel_logp'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(slp[0, :2].cpu().numpy(), g[t + '_logp_rows'], rtol=1e-4, atol=1e-4)
    for k, beams in enumerate(model.done_beams):
        assert len(beams) == int(g['%s_n%d' % (t, k)])
        for j, bm in enumerate(beams):
            assert np.array_equal(bm['seq'].cpu().numpy(), g['%s_%d_%d_seq' % (t, k, j)]), (k, j)
            np.testing.assert_allclose(bm['p'], g['%s_%d_%d_p' % (t, k, j)], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(bm['unaug_p'], g['%s_%d_%d_unaug' % (t, k, j)], rtol=1e-4, atol=1e-4)


MID_CASES = [(n, t, m, kw, over) for n in ('updown', 'aoa', 'transformer')
             for t, m, kw, over in (('mid', False, {}, None), ('midm', True, {}, None), ('midn', True, {'sample_n': 5}, None))]
MID_CASES.append(('transformer', 'tlong', True, {}, (0.0, -60.0, 0.0)))


@pytest.mark.parametrize('name,tag,masked,kw,over', MID_CASES)
def test_beam5_with_beams_ending_mid_sequence_at_config_size_vs_the_reference_itself(name, tag, masked, kw, over):
    """VERDICT r4 weak #3 + missing #6 (tests/golden/beam5_mid.npz, `make_golden.py beam5mid`): (a) beams that end at steps 9-15
    of 20 at V1 = 9488 -- ended beams (their candidates parked at -1000, CaptionModel.py:176-198) and live beams mixed in the
    segmented top-5 over 47 440 candidates for several consecutive steps, which the bimodal beam5_config_size fixture (lengths 0-3
    or 20) never exercised at config size; (b) the Transformer at configs/transformer/transformer.yml size (d = 512, N = 6, h = 8):
    KV caches following the beams by parent pointer (TransformerModel.py:351-362, CaptionModel.py:90-109).  Weights:
    tests/shapes.py:mid_state (a clock feature drives the EOS logit) on both sides; tokens / done beams exact, values <= 1e-4."""
    import shapes
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    g = np.load(os.path.join(GOLDEN, 'beam5_mid.npz'))
    opt = synthetic.updown_opt(drop_prob_lm=0.0) if name == 'updown' else shapes.big_opt(name)
    model = models.setup(opt)
    seed = shapes.BEAM5_MID_SEED[name]
    model.load_state_dict(shapes.mid_state(name, {k: v.shape for k, v in model.state_dict().items()}, seed, *(over or (None, None, None))))
    model = model.to(DEV).eval()
    B, bs = 3, 5
    fc, att = shapes.feats(B, seed=seed)
    am = shapes.ragged_masks(B, seed=seed).to(DEV) if masked else None
    o = {'sample_method': 'beam_search', 'beam_size': bs, 'sample_n': 1}
    o.update(kw)
    with torch.no_grad():
        seq, slp = model(fc.to(DEV), att.to(DEV), am, opt=o, mode='sample')
    t = name + '_' + tag
    assert np.array_equal(seq.cpu().numpy(), g[t + '_seq'])
    sel = slp.gather(2, seq.unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(sel, g[t + '_sel_logp'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(slp[0, :2].cpu().numpy(), g[t + '_logp_rows'], rtol=1e-4, atol=2e-4)
    lens = []
    for k, beams in enumerate(model.done_beams):
        assert len(beams) == int(g['%s_n%d' % (t, k)])
        for j, bm in enumerate(beams):
            assert np.array_equal(bm['seq'].cpu().numpy(), g['%s_%d_%d_seq' % (t, k, j)]), (k, j)
            np.testing.assert_allclose(bm['p'], g['%s_%d_%d_p' % (t, k, j)], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(bm['unaug_p'], g['%s_%d_%d_unaug' % (t, k, j)], rtol=1e-4, atol=1e-4)
            lens.append(int((bm['seq'] > 0).sum()))
    if tag.startswith('mid'):
        assert sum(4 <= l <= 16 for l in lens) >= 5 and len(set(lens)) >= 2, lens      # the fixture does what it is for
    else:
        assert all(l == 20 for l in lens)
