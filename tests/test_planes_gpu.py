"""The decode GEMMs take PRODUCER-WRITTEN bf16x3 "A planes" and stage them by LDS-DMA (gemm_lc.hip loader / consumer kernel).
Checked here, through the C ABI: the conversion entry point and every fused producer against the numpy restatement of
the layout (oracle/planes.py, bit-exact); the planes GEMM against fp64 (fp32-grade error) and against the in-kernel-split
GEMM; a rollout with planes against the same rollout without."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import planes as PL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def mods():
    from imagecaptioning.pytorch_amd import ops, _lib
    return ops, _lib


def wide(shape, g, scale=1.0):
    x = torch.randn(shape, generator=g)
    return (x * torch.exp(1.5 * torch.randn(shape, generator=g)) * scale).float()


@pytest.mark.parametrize('M,K', [(60, 1000), (64, 32), (1, 7), (33, 4000), (10, 513)])
def test_planes_from_f32_matches_layout_restatement(dev, M, K):
    ops, _ = mods()
    g = torch.Generator().manual_seed(M + K)
    x = wide((M, K), g)
    pl = ops.planes_from_f32(x.to(dev))
    torch.cuda.synchronize()
    assert np.array_equal(pl.cpu().numpy(), PL.planes_from_f32(x.numpy()))


# (M, N, [K per segment], b_layout, deferred)
SHAPES = [(60, 4000, [1000, 1000, 1000], 0, True),      # LSTM gate GEMM of the SCST step
          (60, 9488, [1000], 0, True),                   # vocabulary projection
          (60, 512, [1000], 0, True),                    # h2att
          (50, 3000, [4000], 1, True),                   # dX = dG [W_ih | W_hh] of the language LSTM
          (50, 2000, [4000], 1, True),                   # ... of the attention LSTM
          (10, 4000, [1000, 1000, 1000], 0, True),       # M <= 32 instance
          (33, 130, [40, 8], 0, False),                  # tiny, unaligned N, epilogue path
          (64, 257, [96], 1, False),
          (7, 64, [32], 0, False)]


@pytest.mark.parametrize('M,N,Ks,bl,defer', SHAPES)
def test_gemm_with_planes_is_fp32_grade_and_agrees_with_in_kernel_split(dev, M, N, Ks, bl, defer):
    ops, _ = mods()
    g = torch.Generator().manual_seed(N + sum(Ks))
    As = [wide((M, K), g, 0.3).to(dev) for K in Ks]
    Bs = [(torch.randn((K, N) if bl else (N, K), generator=g) * 0.05).to(dev) for K in Ks]
    segs = [(A, K, B, N if bl else K, K, 1) for A, B, K in zip(As, Bs, Ks)]
    planes = [ops.planes_from_f32(A) for A in As]
    ws = ops.Workspace(dev, 32 << 20)
    bias = torch.randn(N, generator=g).to(dev)

    def run(a_planes):
        out = torch.zeros(M, N, device=dev)
        splits = ops.gemm(segs, M, N, out, b_layout=bl, ws=ws, defer_reduce=defer, bias=None if defer else bias,
                          a_planes=a_planes)
        if defer:
            out = ws.slabs[:splits * M * N].view(splits, M, N).sum(0)
        return out.clone(), splits

    o_pl, s_pl = run(planes)
    o_ref, s_ref = run(None)
    ref = sum(A.double() @ (B.double() if bl else B.double().t()) for A, B in zip(As, Bs))
    if not defer:
        ref = ref + bias.double()
    mag = sum(A.double().abs() @ (B.double().abs() if bl else B.double().abs().t()) for A, B in zip(As, Bs)) + 1e-30
    # same exact bf16x3 products, another summation order (K slices / chunk order): fp32 accumulation noise only
    assert float(((o_pl.double() - ref).abs() / mag).max()) < 1e-6
    assert float(((o_pl.double() - o_ref.double()).abs() / mag).max()) < 1e-6


def test_producers_write_the_planes_of_their_outputs(dev):
    """LSTM cell (h, h_drop), attention (ctx), embedding (x), cell backward (d_gates): planes == planes_from_f32(fp32 output)."""
    ops, _lib = mods()
    lib, sp = _lib.lib, _lib.stream_ptr
    g = torch.Generator().manual_seed(5)
    N, R, E, A, K, B, n, V1 = 60, 1000, 1000, 512, 36, 12, 5, 200
    z = lambda *s: torch.zeros(*s, device=dev)       # noqa: E731
    zb = lambda k: torch.zeros(int(lib.capmi_planes_bytes(k)), dtype=torch.uint8, device=dev)      # noqa: E731

    def same(pl, x):
        torch.cuda.synchronize()
        assert np.array_equal(pl.cpu().numpy(), PL.planes_from_f32(x.cpu().numpy()))

    # forward cell over 3 slabs
    slabs = (torch.randn(3, N, 4 * R, generator=g)).to(dev)
    b1, b2 = torch.randn(4 * R, generator=g).to(dev), torch.randn(4 * R, generator=g).to(dev)
    cp = torch.randn(N, R, generator=g).to(dev)
    mask = ((torch.rand(N, R, generator=g) < 0.5).float() * 2).to(dev)
    h, c, ga, hd, h0, c0, ga0, hd0 = (z(N, R), z(N, R), z(N, 4 * R), z(N, R), z(N, R), z(N, R), z(N, 4 * R), z(N, R))
    pl_h, pl_hd = zb(R), zb(R)
    _lib.check(lib.capmi_lstm_cell_fwd_pl(slabs.data_ptr(), 3, b1.data_ptr(), b2.data_ptr(), None, 1, None, cp.data_ptr(),
                                          h.data_ptr(), c.data_ptr(), ga.data_ptr(), mask.data_ptr(), hd.data_ptr(), N, R,
                                          pl_h.data_ptr(), pl_hd.data_ptr(), sp()), 'cell_pl')
    _lib.check(lib.capmi_lstm_cell_fwd(slabs.data_ptr(), 3, b1.data_ptr(), b2.data_ptr(), None, 1, None, cp.data_ptr(),
                                       h0.data_ptr(), c0.data_ptr(), ga0.data_ptr(), mask.data_ptr(), hd0.data_ptr(), N, R,
                                       sp()), 'cell')
    for a, b in ((h, h0), (c, c0), (ga, ga0), (hd, hd0)):    # the 16-byte cell vs the scalar cell (fma contraction may differ)
        assert float((a - b).abs().max()) < 1e-6
    same(pl_h, h); same(pl_hd, hd)

    # embedding
    Emb = torch.randn(V1, E, generator=g).to(dev)
    it = torch.randint(0, V1, (N,), generator=g).to(dev)
    x, pl_x = z(N, E), zb(E)
    _lib.check(lib.capmi_embed_fwd_pl(it.data_ptr(), 1, None, Emb.data_ptr(), None, x.data_ptr(), N, E, 1, pl_x.data_ptr(), sp()),
               'embed_pl')
    same(pl_x, x)

    # attention (rows of an image grouped: B * n = 60)
    att_h = torch.randn(2, N, A, generator=g).to(dev)
    p_att, att = torch.randn(B, K, A, generator=g).to(dev), torch.randn(B, K, R, generator=g).to(dev)
    w, bb, hb = torch.randn(A, generator=g).to(dev), torch.randn(1, generator=g).to(dev), torch.randn(A, generator=g).to(dev)
    ctx, alpha, aho, pl_ctx = z(N, R), z(N, K), z(N, A), zb(R)
    _lib.check(lib.capmi_attention_fwd_partial_pl(att_h.data_ptr(), 2, N * A, hb.data_ptr(), aho.data_ptr(), p_att.data_ptr(),
                                                  att.data_ptr(), None, w.data_ptr(), bb.data_ptr(), ctx.data_ptr(),
                                                  alpha.data_ptr(), B, n, K, A, R, None, N, pl_ctx.data_ptr(), sp()), 'att_pl')
    same(pl_ctx, ctx)
    row_img = torch.arange(N, dtype=torch.int32).remainder(B).to(dev)       # ragged grouping: one row per workgroup
    ctx2, pl_ctx2 = z(N, R), zb(R)
    _lib.check(lib.capmi_attention_fwd_partial_pl(att_h.data_ptr(), 2, N * A, hb.data_ptr(), aho.data_ptr(), p_att.data_ptr(),
                                                  att.data_ptr(), None, w.data_ptr(), bb.data_ptr(), ctx2.data_ptr(),
                                                  alpha.data_ptr(), B, n, K, A, R, row_img.data_ptr(), N, pl_ctx2.data_ptr(),
                                                  sp()), 'att_pl_rows')
    same(pl_ctx2, ctx2)

    # backward cell
    Nb = 50
    dh = torch.randn(Nb, R, generator=g).to(dev)
    gates = torch.rand(Nb, 4 * R, generator=g).to(dev)
    cprev, cnew = torch.randn(Nb, R, generator=g).to(dev), torch.randn(Nb, R, generator=g).to(dev)
    dg, dcp, pl_dg = z(Nb, 4 * R), z(Nb, R), zb(4 * R)
    _lib.check(lib.capmi_lstm_cell_bwd_partial_pl(dh.data_ptr(), R, None, None, 0, 1, 0, None, 0, 1, 0, None, gates.data_ptr(),
                                                  cprev.data_ptr(), cnew.data_ptr(), dg.data_ptr(), dcp.data_ptr(), Nb, R,
                                                  pl_dg.data_ptr(), sp()), 'cell_bwd_pl')
    same(pl_dg, dg)


def test_rollout_with_planes_agrees_with_rollout_without(dev, monkeypatch):
    """SCST-shaped rollout at the BASELINE sizes, forward + BPTT, with dropout masks and Gumbel noise injected: the planes are
    another delivery of the same operands, so the tokens coincide and log-probs / gradients agree to fp32 accumulation noise
    (the planes-off run is teacher-forced on the planes-on tokens so that a near-tie cannot fork the two)."""
    from imagecaptioning.pytorch_amd import updown_engine as E
    from shapes import full_size_params
    torch.manual_seed(0)
    B, n, K, R, Em, A, V1, L = 10, 5, 36, 1000, 1000, 512, 9488, 20
    P = {k: v.to(dev).contiguous() for k, v in full_size_params(seed=3).items()}
    fc = torch.randn(B, 2048, device=dev).clamp_min(0)
    att = torch.randn(B, K, 2048, device=dev).clamp_min(0)
    pr = E.prepare(P, fc, att, None)
    N = B * n
    gum = torch.rand(L, N, V1, device=dev).clamp_min(1e-12).log().neg().log().neg()
    drop_xt = (torch.rand(L, N, Em, device=dev) < 0.5).float() * 2
    drop_out = (torch.rand(L, N, R, device=dev) < 0.5).float() * 2
    gsel = -torch.rand(N, L, 1, device=dev)
    out = {}
    seq_on = None
    for flag in ('1', '0', 'free'):
        monkeypatch.setenv('CAPMI_PLANES', '1' if flag == '1' else '0')
        kw = dict(mode='sample', gumbel=gum) if flag != '0' else dict(mode='forced', forced=seq_on)
        ro = E.Rollout(P, pr, n=n, T=L, drop_xt=drop_xt, drop_out=drop_out, **kw)
        assert (ro.r.planes is not None) == (flag == '1')
        seq, slp = ro.run()
        if flag == 'free':
            torch.cuda.synchronize()
            assert float((seq != seq_on).float().mean()) < 0.02       # free-running without planes: same tokens (near-ties aside)
            continue
        if flag == '1':
            seq_on = seq.clone()
        grads = {k: torch.zeros_like(P[k]) for k in E.PARAM_KEYS}
        gsl = torch.zeros_like(slp)
        gsl.scatter_(2, seq_on.unsqueeze(-1), gsel)
        d = ro.backward(gsl, grads)
        torch.cuda.synchronize()
        out[flag] = (seq.clone(), slp.gather(2, seq_on.unsqueeze(-1)).clone(), {k: v.clone() for k, v in grads.items()},
                     [t.clone() for t in d])
    a, b = out['1'], out['0']
    assert torch.equal(a[0], b[0])
    assert float((a[1] - b[1]).abs().max()) < 2e-5
    # (alpha_net.bias is left out: the softmax is shift invariant, its gradient is rounding noise around an exact 0)
    errs = {k: float((a[2][k] - b[2][k]).abs().max()) / (float(b[2][k].abs().max()) + 1e-30) for k in a[2]
            if k != 'core.attention.alpha_net.bias'}
    errs.update({'d%d' % i: float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30) for i, (x, y) in enumerate(zip(a[3], b[3]))})
    print('relative gradient differences planes on / off:', {k: '%.1e' % v for k, v in errs.items() if v > 1e-5})
    # fp32 accumulation-order noise through 20 steps of BPTT (the oracle comparisons of test_full_size_parity_gpu.py allow 1e-3)
    assert max(errs.values()) < 5e-4, {k: v for k, v in errs.items() if v > 1e-5}
