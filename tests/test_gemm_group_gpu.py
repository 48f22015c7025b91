"""capmi_gemm_group_tn (r6): n independent weight-gradient GEMMs dW_i = dY_i^T X_i in one persistent launch of the 256 x 128 bf16x3
kernel (+ one reduction launch for the K-sliced tail) -- the nn.Linear / nn.LSTMCell weight gradients autograd produces one by one
behind `loss.backward()` (reference tools/train.py:193).  Checked against fp64, and BIT FOR BIT against capmi_gemm_f32 on each item
with the same K split (tiles of the full rounds: whole-K; tail tiles: the split reported in splits_used)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _ops():
    from imagecaptioning.pytorch_amd import ops
    return ops


def _items(shapes, dev, seed=0, wide_range=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for (M, N, K) in shapes:
        dy = torch.randn(K, M, generator=g)
        if wide_range:
            dy = dy * torch.exp(2.0 * torch.randn(dy.shape, generator=g))
        x = torch.randn(K, N, generator=g)
        out.append((dy.to(dev), x.to(dev), torch.full((M, N), float('nan'), device=dev), False))
    return out


SETS = {
    # the UpDown SCST step's seven LSTM / h2att weight gradients + dW_logit (R = E = 1000, A = 512, T*N = 1000 rows)
    'scst': [(4000, 1000, 1000)] * 6 + [(512, 1000, 1000), (9488, 1000, 1000)],
    # one Transformer decoder layer at bs64 x 5, T = 21: eight d x d projections + the two FFN matrices
    'txe_layer': [(512, 512, 6720)] * 8 + [(2048, 512, 6720), (512, 2048, 6720)],
    # fewer tiles than CUs: everything is tail (K-sliced).  (M * N >= 256 K: the single launches the bitwise test compares with take
    # the bf16x3 kernel too -- smaller products go to the exact-fp32 tile kernel there)
    'small': [(512, 512, 2304), (1024, 256, 640)],
    # ragged edges, mixed K, an item the fat kernel cannot take (N % 4 != 0 -> capmi_gemm_f32 behind the group)
    'ragged': [(260, 132, 100), (1028, 516, 1000), (300, 200, 36), (128, 130, 64), (4, 4, 4)],
    # K not a multiple of 4 (r6: T * N rows of a rollout -- 50 rows x an odd number of steps): the last K tile's quads are cut
    'odd_k': [(4000, 1000, 950), (512, 1000, 1050), (1000, 512, 1050), (256, 128, 37), (128, 128, 3), (512, 512, 2302)],
    # more items than one table holds (two launches, longest K first)
    'many': [(256, 128, 64 * (1 + i % 5)) for i in range(60)],
}


@pytest.mark.parametrize('name', sorted(SETS))
def test_group_matches_fp64(dev, name):
    ops = _ops()
    items = _items(SETS[name], dev, seed=len(name), wide_range=(name == 'scst'))
    ops.gemm_group_tn(items)
    for dy, x, out, _ in items:
        ref = dy.double().t() @ x.double()
        mag = dy.double().abs().t() @ x.double().abs()
        e_ours = float(((out.double() - ref).abs() / (mag + 1e-30)).max())
        e_fp32 = float((((dy.t() @ x).double() - ref).abs() / (mag + 1e-30)).max())
        assert torch.isfinite(out).all()
        # (the dropped cross terms of the 3-way split are <= 3 * 2^-24 |a||b| = 1.8e-7 PER PRODUCT: with a handful of products -- K = 3 --
        #  nothing averages, the bound itself is what is seen)
        assert e_ours <= 1.5 * e_fp32 + (1e-7 if dy.shape[0] >= 32 else 4e-7), (name, tuple(out.shape), e_ours, e_fp32)


@pytest.mark.parametrize('name', ['scst', 'txe_layer', 'small', 'ragged', 'many', 'odd_k'])
def test_group_column_sums_ride_along(dev, name):
    """capmi_group_gemm.colsum: the bias gradient (column sums of dY) taken by the staging waves of an item's first column tiles --
    whole-K units, K-sliced tail units (pieces + the reduction launch) and items that fall back to capmi_gemm_f32; the products
    themselves are unchanged by it, bit for bit"""
    ops = _ops()
    plain = _items(SETS[name], dev, seed=23)
    ops.gemm_group_tn(plain)
    items = [(dy, x, torch.full_like(out, float('nan')), False, None, 0, torch.full((dy.shape[1],), float('nan'), device=dev))
             for dy, x, out, _ in plain]
    ops.gemm_group_tn(items)
    for (dy, x, out, _, _, _, cs), ref in zip(items, plain):
        assert torch.equal(out, ref[2])
        want = dy.double().sum(0)
        mag = dy.double().abs().sum(0)
        assert torch.isfinite(cs).all()
        assert float(((cs.double() - want).abs() / (mag + 1e-30)).max()) < 2e-6, (name, tuple(dy.shape))
    again = [it[:6] + (torch.empty_like(it[6]),) for it in items]
    ops.gemm_group_tn(again)
    for a, b in zip(items, again):
        assert torch.equal(a[6], b[6])                     # deterministic: fixed summation order, no atomics


def test_layernorm_bwd_parts(dev):
    """capmi_layernorm_bwd_parts: dx bit-identical to capmi_layernorm_bwd; the column sums of the per-wave partial rows are d_a / d_b"""
    from imagecaptioning.pytorch_amd._lib import lib, ptr, check, stream_ptr
    g = torch.Generator().manual_seed(3)
    for M, D in ((6720, 512), (37, 512), (360, 1024), (5, 260)):
        dy, x, a = (torch.randn(s, generator=g).to(dev) for s in ((M, D), (M, D), (D,)))
        mean = x.mean(1).contiguous()
        inv = (1.0 / (x.std(1) + 1e-6)).contiguous()
        dx1, dx2, gs = torch.zeros(M, D, device=dev), torch.zeros(M, D, device=dev), torch.empty(M, D, device=dev)
        check(lib.capmi_layernorm_bwd(ptr(dy), ptr(x), ptr(a), ptr(mean), ptr(inv), ptr(dx1), 1, ptr(gs), M, D, 1e-6, stream_ptr()), 'ln')
        W = int(lib.capmi_layernorm_bwd_parts_rows(M))
        parts = torch.full((2, W, D), float('nan'), device=dev)
        check(lib.capmi_layernorm_bwd_parts(ptr(dy), ptr(x), ptr(a), ptr(mean), ptr(inv), ptr(dx2), 1, parts[0].data_ptr(), parts[1].data_ptr(),
                                            M, D, 1e-6, stream_ptr()), 'ln parts')
        assert torch.equal(dx1, dx2)
        for got, want in ((parts[0], gs), (parts[1], dy)):
            ref = want.double().sum(0)
            mag = want.double().abs().sum(0)
            assert float(((got.double().sum(0) - ref).abs() / (mag + 1e-30)).max()) < 2e-6, (M, D)


def test_relu_scale_bwd_is_the_masked_jacobian(dev):
    """capmi_relu_scale_bwd (y_ref = relu(pre) * mask, the mask not read) == capmi_relu_mask_bwd with the mask, bit for bit"""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    pre = torch.randn(333, 2048, generator=g).to(dev)
    dy = torch.randn(333, 2048, generator=g).to(dev)
    for p in (0.1, 0.3, 0.5):
        mask = ops.dropout_mask((333, 2048), p, 1234, 5 << 36, dev)
        assert hasattr(mask, '_capmi_scale') and float(mask.max()) == mask._capmi_scale[0]
        y = torch.relu(pre) * mask
        fast = ops.relu_mask_bwd(dy, y, mask)
        plain = mask.clone()                                   # (no keep-scale attribute: the three-operand kernel)
        slow = ops.relu_mask_bwd(dy, y, plain)
        assert torch.equal(fast, slow)
        assert torch.equal(fast, dy * mask * (pre > 0))
        # a mask edited in place after the kernel made it no longer vouches for its keep-scale: the three-operand kernel runs
        edited = ops.dropout_mask((333, 2048), p, 99, 7 << 36, dev)
        edited.mul_(0.5)
        y2 = torch.relu(pre) * edited
        assert torch.equal(ops.relu_mask_bwd(dy, y2, edited), dy * edited * (pre > 0))


def test_group_accumulates(dev):
    ops = _ops()
    items = _items([(512, 512, 800), (1024, 256, 800), (260, 132, 100)], dev, seed=5)
    base = [torch.randn_like(o) for _, _, o, _ in items]
    items = [(dy, x, b.clone(), True) for (dy, x, _, _), b in zip(items, base)]
    ops.gemm_group_tn(items)
    for (dy, x, out, _), b in zip(items, base):
        ref = b.double() + dy.double().t() @ x.double()
        assert float((out.double() - ref).abs().max() / ref.abs().max()) < 3e-6


@pytest.mark.parametrize('name', ['scst', 'txe_layer', 'small'])
def test_group_is_bitwise_the_single_launch_with_the_same_k_split(dev, name):
    """every output element equals capmi_gemm_f32's on that item with splits = 1 (tiles inside the full rounds) or with the tail's
    split; items whose splits_used is 1 equal the whole-K launch everywhere"""
    ops = _ops()
    items = _items(SETS[name], dev, seed=17)
    used = ops.gemm_group_tn(items)
    for (dy, x, out, _), sp in zip(items, used):
        K, M = dy.shape
        N = x.shape[1]
        whole = torch.empty(M, N, device=dev)
        ops.gemm([(dy, M, x, N, K, 1)], M, N, whole, a_layout=1, b_layout=1, splits=1)
        if sp == 1:
            assert torch.equal(out, whole), (name, M, N, K)
            continue
        cut = torch.empty(M, N, device=dev)
        ws = ops.Workspace(dev, floats=sp * M * N + 1024)
        got = ops.gemm([(dy, M, x, N, K, 1)], M, N, cut, a_layout=1, b_layout=1, splits=sp, ws=ws)
        del ws
        assert got == sp
        same = (out == whole) | (out == cut)
        assert bool(same.all()), (name, M, N, K, sp, int((~same).sum()))
        # ... and tile by tile: a 256 x 128 tile is entirely one or the other
        for m0 in range(0, M, 256):
            for n0 in range(0, N, 128):
                t, w, c = out[m0:m0 + 256, n0:n0 + 128], whole[m0:m0 + 256, n0:n0 + 128], cut[m0:m0 + 256, n0:n0 + 128]
                assert torch.equal(t, w) or torch.equal(t, c)


def test_group_switch_off_is_the_single_launches(dev, monkeypatch):
    """CAPMI_GEMM_GROUP is read once per process: here only the documented fallback of ineligible items is exercised (every item of
    this list is ineligible) and must agree with ops.gemm"""
    ops = _ops()
    items = _items([(130, 70, 50), (66, 258, 36)], dev, seed=2)
    used = ops.gemm_group_tn(items)
    assert used == [-1, -1]
    for dy, x, out, _ in items:
        K, M = dy.shape
        N = x.shape[1]
        ref = torch.empty(M, N, device=dev)
        ops.gemm([(dy, M, x, N, K, 1)], M, N, ref, a_layout=1, b_layout=1, splits=1)
        assert torch.equal(out, ref)
