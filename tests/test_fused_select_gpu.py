"""The fused select + GEMM launch of the decode step (round 4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gate_gemm_part_inside_the_select_launch_agrees_with_the_plain_step(monkeypatch):
    """r4: the K segments of the attention-LSTM gate GEMM fed by h_lang(t) / h_att(t) are computed by the idle workgroups of step t's
    select launch (capmi_logsoftmax_select_partial_gemm) and meet the token-embedding segment in the LSTM cell.  Same operands, another
    summation order: free-running tokens coincide, log-probs / gradients agree to fp32 accumulation noise; with early exit and a
    model that ends its captions nothing is left behind."""
    dev = torch.device('cuda:0')
    from imagecaptioning.pytorch_amd import updown_engine as E
    from shapes import full_size_params
    torch.manual_seed(1)
    B, n, K, R, Em, V1, L = 10, 5, 36, 1000, 1000, 9488, 20
    P = {k: v.to(dev).contiguous() for k, v in full_size_params(seed=4).items()}
    fc = torch.randn(B, 2048, device=dev).clamp_min(0)
    att = torch.randn(B, K, 2048, device=dev).clamp_min(0)
    pr = E.prepare(P, fc, att, None)
    N = B * n
    gum = torch.rand(L, N, V1, device=dev).clamp_min(1e-12).log().neg().log().neg()
    drop_xt = (torch.rand(L, N, Em, device=dev) < 0.5).float() * 2
    drop_out = (torch.rand(L, N, R, device=dev) < 0.5).float() * 2
    gsel = -torch.rand(N, L, 1, device=dev)
    out, seq_on = {}, None
    for flag in ('1', '0'):
        monkeypatch.setenv('CAPMI_FUSED_SELECT', flag)
        kw = dict(mode='sample', gumbel=gum) if flag == '1' else dict(mode='forced', forced=seq_on)
        ro = E.Rollout(P, pr, n=n, T=L, drop_xt=drop_xt, drop_out=drop_out, **kw)
        assert (ro.r.pre_partial is not None) == (flag == '1')
        seq, slp = ro.run()
        if flag == '1':
            seq_on = seq.clone()
        grads = {k: torch.zeros_like(P[k]) for k in E.PARAM_KEYS}
        gsl = torch.zeros_like(slp)
        gsl.scatter_(2, seq_on.unsqueeze(-1), gsel)
        ro.backward(gsl, grads)
        torch.cuda.synchronize()
        out[flag] = (seq.clone(), slp.gather(2, seq_on.unsqueeze(-1)).clone(), {k: v.clone() for k, v in grads.items()})
    a, b = out['1'], out['0']
    assert torch.equal(a[0], b[0])
    assert float((a[1] - b[1]).abs().max()) < 2e-5
    errs = {k: float((a[2][k] - b[2][k]).abs().max()) / (float(b[2][k].abs().max()) + 1e-30) for k in a[2]
            if k != 'core.attention.alpha_net.bias'}
    assert max(errs.values()) < 5e-4, {k: v for k, v in errs.items() if v > 1e-5}
    # free-running with one gate GEMM per step: the same tokens (near-ties aside)
    monkeypatch.setenv('CAPMI_FUSED_SELECT', '0')
    seq0, _ = E.Rollout(P, pr, n=n, T=L, mode='sample', gumbel=gum, drop_xt=drop_xt, drop_out=drop_out).run()
    torch.cuda.synchronize()
    assert float((seq0 != seq_on).float().mean()) < 0.02




@pytest.mark.parametrize('mode,top_k,top_p', [(1, 0, 0.0), (0, 0, 0.0), (1, 7, 0.0), (1, 0, 0.8)])
def test_fused_launch_is_bit_identical_to_the_two_launches(mode, top_k, top_p):
    """capmi_logsoftmax_select_partial_gemm == capmi_gemm_f32 (loader / consumer kernel, slabs) followed by capmi_logsoftmax_select_partial
    on the same inputs: the two halves of the fused grid run the same bodies, so tokens, dense log-probs, the embedded next input and
    the GEMM's K-slice slabs are the same BITS; a GEMM that cannot ride along (N rows + its workgroups > 256) falls back to the two
    launches inside the call."""
    import ctypes as C
    from imagecaptioning.pytorch_amd import ops, _lib
    from imagecaptioning.pytorch_amd._lib import lib, ptr, stream_ptr, check
    dev = torch.device('cuda:0')

    class NextEmbed(C.Structure):
        _fields_ = [('E', C.c_void_p), ('mask', C.c_void_p), ('x', C.c_void_p), ('it_save', C.c_void_p), ('Edim', C.c_int),
                    ('relu', C.c_int), ('x_planes', C.c_void_p), ('alive', C.c_void_p)]
    g = torch.Generator().manual_seed(3)
    N, V1, R, E, L, step = 60, 9488, 1000, 1000, 20, 3
    logit_slabs = (torch.randn(3, N, V1, generator=g) * 0.7).to(dev)
    bias = torch.randn(V1, generator=g).to(dev)
    emb = torch.randn(V1, E, generator=g).to(dev)
    mask = ((torch.rand(N, E, generator=g) < 0.5).float() * 2).to(dev)
    h1, h2 = torch.randn(N, R, generator=g).to(dev), torch.randn(N, R, generator=g).to(dev)
    W1, W2 = (torch.randn(4 * R, 3 * R, generator=g) * 0.03).to(dev), (torch.randn(4 * R, R, generator=g) * 0.03).to(dev)
    pl1, pl2 = ops.planes_from_f32(h1), ops.planes_from_f32(h2)

    def run(fused, splits_hint):
        ws = ops.Workspace(dev, floats=8 * 64 * 4 * R)
        seq = torch.zeros(N, L, dtype=torch.long, device=dev)
        it = torch.zeros(N, dtype=torch.long, device=dev)
        unf = torch.ones(N, dtype=torch.uint8, device=dev)
        slp = torch.zeros(N, L, V1, device=dev)
        sel = torch.zeros(N, L, device=dev)
        live = torch.zeros(N, L, dtype=torch.uint8, device=dev)
        x_next = torch.zeros(N, E, device=dev)
        x_pl = torch.zeros(int(lib.capmi_planes_bytes(E)), dtype=torch.uint8, device=dev)
        ne = NextEmbed()                   # capmi.h capmi_next_embed
        ne.E, ne.mask, ne.x, ne.Edim, ne.relu, ne.x_planes = emb.data_ptr(), mask.data_ptr(), x_next.data_ptr(), E, 1, x_pl.data_ptr()
        d = _lib.GemmDesc()
        d.nseg = 2
        d.seg[0].A, d.seg[0].lda, d.seg[0].B, d.seg[0].ldb, d.seg[0].K, d.seg[0].a_row_div = h1.data_ptr(), R, W1.data_ptr(), 3 * R, R, 1
        d.seg[1].A, d.seg[1].lda, d.seg[1].B, d.seg[1].ldb, d.seg[1].K, d.seg[1].a_row_div = h2.data_ptr(), R, W2.data_ptr(), R, R, 1
        d.a_planes[0], d.a_planes[1] = pl1.data_ptr(), pl2.data_ptr()
        d.M, d.N, d.C, d.ldc = N, 4 * R, ws.buf.data_ptr(), 4 * R
        d.partial, d.partial_capacity, d.splits, d.defer_reduce = ws.buf.data_ptr(), ws.capacity, splits_hint, 1
        flt = _lib.SampleFilter()
        flt.top_k, flt.top_p = top_k, top_p
        sel_args = (logit_slabs.data_ptr(), 3, N * V1, bias.data_ptr(), N, V1, step, L, mode, None, 1.0, None, 1234, None, 0, 0,
                    seq.data_ptr(), L, it.data_ptr(), unf.data_ptr(), slp.data_ptr(), sel.data_ptr(), live.data_ptr(), C.byref(ne),
                    C.byref(flt) if (top_k or top_p) else None)
        if fused:
            check(lib.capmi_logsoftmax_select_partial_gemm(*sel_args, C.byref(d), stream_ptr()), 'fused')
        else:
            check(lib.capmi_gemm_f32(C.byref(d), stream_ptr()), 'gemm')
            check(lib.capmi_logsoftmax_select_partial(*sel_args, stream_ptr()), 'select')
        torch.cuda.synchronize()
        sp = int(d.splits_used)
        return seq[:, step].clone(), slp[:, step].clone(), sel[:, step].clone(), x_next.clone(), x_pl.clone(), sp, ws.slabs[:sp * N * 4 * R].clone()

    for hint in (6, 8):                      # 6: 60 + 192 workgroups ride in one grid; 8: 60 + 256 do not -> two launches inside the call
        a, b = run(True, hint), run(False, hint)
        assert a[5] == b[5] == hint
        for x, y in zip(a[:5] + a[6:], b[:5] + b[6:]):
            assert torch.equal(x, y)
    assert int(a[0].max()) > 0


def test_raw_logit_rollouts_and_a_margin_structure_loss_vs_oracle():
    """VERDICT r3 missing #7 -- AttModel._sample(output_logsoftmax=0) (AttModel.py:171-175, 265; loss_wrapper.py:31-37): the rollout
    stores the LOGITS (CAPMI_SELECT_RAW) and its backward takes the loss gradient as d(logits).  Against oracle/att_lstm.py with the
    same Gumbel noise: tokens equal, stored rows == the oracle's logits (<= 2e-5; zero rows after the end), raw - logsumexp(raw) ==
    the log-prob rollout's rows; the 'max_margin' structure loss (sparse gradient route) and every parameter gradient match the
    oracle's autograd (<= 1e-3 relative)."""
    import argparse
    from oracle import att_lstm as O
    from test_model_api_gpu import golden_model, DEV
    from imagecaptioning.pytorch_amd.captioning.modules import losses as Lm
    z, model = golden_model(False)
    model.eval()                                   # no dropout: the same masks (none) on both sides
    fc, att = torch.from_numpy(z['fc']), torch.from_numpy(z['att'])
    B, n, L = fc.shape[0], 3, model.seq_length
    N, V1 = B * n, model.vocab_size + 1
    g = torch.Generator().manual_seed(9)
    gum = -torch.log(-torch.log(torch.rand(L, N, V1, generator=g).clamp_min(1e-20)))
    scores = torch.rand(N, generator=g)
    o = {'sample_method': 'sample', 'sample_n': n, '_gumbel': gum.to(DEV)}
    with torch.no_grad():
        seq_lp, lp = model(fc.to(DEV), att.to(DEV), None, opt=dict(o, output_logsoftmax=1), mode='sample')
    seq, raw = model(fc.to(DEV), att.to(DEV), None, opt=dict(o, output_logsoftmax=0), mode='sample')
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    seq_o, raw_o = O.rollout(P, fc, att, None, method='sample', sample_n=n, max_len=L, gumbel=gum, output_logsoftmax=False)
    assert torch.equal(seq.cpu(), seq_o) and torch.equal(seq_lp.cpu(), seq_o)
    assert float((raw.detach().cpu() - raw_o.detach()).abs().max()) < 2e-5
    live = torch.cat([torch.ones(N, 1, dtype=torch.bool), seq_o[:, :-1] > 0], 1)
    assert float((torch.log_softmax(raw.detach().cpu(), 2) - lp.cpu())[live].abs().max()) < 2e-5
    assert float(raw.detach().cpu()[~live].abs().max() if (~live).any() else 0.0) == 0.0
    opt = argparse.Namespace(structure_loss_type='max_margin', train_sample_n=n, entropy_reward_weight=0, self_cider_reward_weight=0,
                             cider_reward_weight=1)
    saved = Lm.get_scores
    Lm.get_scores = lambda data_gts, gen_result, op, as_tensor=False: scores.clone().to(gen_result.device)
    try:
        loss = Lm.StructureLosses(opt)(raw, seq, [None] * B)['loss']
        loss_o = Lm.StructureLosses(opt)(raw_o, seq_o, [None] * B)['loss']
    finally:
        Lm.get_scores = saved
    assert abs(loss.item() - loss_o.item()) < 1e-5 and loss.item() > 0
    model.zero_grad()
    loss.backward()
    loss_o.backward()
    for k, p in model.named_parameters():
        ref = P[k].grad
        if k.endswith('alpha_net.bias') or ref is None:
            continue
        err = float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        assert err < 1e-3, (k, err)


def test_raw_logit_rollout_at_baseline_size_register_resident_select():
    """the same at R = E = 1000, V1 = 9488 (the register-resident select, inside the fused select + GEMM launch): tokens equal the
    log-prob rollout's, log_softmax(stored logits) == its rows, the selected entries line up"""
    from imagecaptioning.pytorch_amd import updown_engine as E
    from shapes import full_size_params
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    B, n, K, L, V1 = 4, 5, 36, 20, 9488
    P = {k: v.to(dev).contiguous() for k, v in full_size_params(seed=8).items()}
    P['logit.bias'] = P['logit.bias'].clone()
    P['logit.bias'][0] += 3.0                       # some captions end: rows of zeros behind the end in both rollouts
    fc = torch.randn(B, 2048, device=dev).clamp_min(0)
    att = torch.randn(B, K, 2048, device=dev).clamp_min(0)
    pr = E.prepare(P, fc, att, None)
    gum = torch.rand(L, B * n, V1, device=dev).clamp_min(1e-12).log().neg().log().neg()
    a = E.Rollout(P, pr, n=n, T=L, mode='sample', gumbel=gum)
    seq_a, lp = a.run()
    b = E.Rollout(P, pr, n=n, T=L, mode='sample', gumbel=gum, raw_logits=True)
    seq_b, raw = b.run()
    torch.cuda.synchronize()
    assert torch.equal(seq_a, seq_b)
    live = torch.cat([torch.ones(B * n, 1, dtype=torch.bool, device=dev), seq_a[:, :-1] > 0], 1)
    assert float((torch.log_softmax(raw, 2) - lp)[live].abs().max()) < 2e-5
    assert float(raw[~live].abs().max()) == 0.0 if bool((~live).any()) else True
    sel_raw = raw.gather(2, seq_b.unsqueeze(2)).squeeze(2)
    assert torch.equal(b.sel_logp, sel_raw * live.float())
