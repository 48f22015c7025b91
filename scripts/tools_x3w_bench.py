#!/usr/bin/env python3
"""Fat GEMM (bf16x3) on 128 x 128 vs 256 x 128 tiles: time, TFLOP/s and error against an fp64 product, for the shapes of the XE
steps (Transformer FFN / projections, UpDown per-step gate GEMMs at 320 rows, time-batched dW / dX, the vocabulary projection)
and for edge shapes (M, N, K off every tile multiple; every operand layout; every epilogue operand).

    CAPMI_X3_TILE=128 python scripts/tools_x3w_bench.py      # narrow tiles only
    CAPMI_X3_TILE=256 python scripts/tools_x3w_bench.py      # wide tiles wherever the bf16x3 path applies
    python scripts/tools_x3w_bench.py                        # the planner's choice (CAPMI_X3W_COST)
One line per shape: us per launch (best of 3 x 20), TFLOP/s, max |err| / max |ref|."""
import os
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')
quick = '--check' in sys.argv
short = '--short' in sys.argv          # seven shapes, no edge sweep: variant A/Bs of the wide kernel (CAPMI_X3W_NSW / CAPMI_X3W_PRIO)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


ws = ops.Workspace(dev, 256 << 20)
g = torch.Generator(device='cpu').manual_seed(5)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def run(name, M, N, Ks, al, bl, epi=False, time_it=True):
    """Ks: K of each segment (several: the [h | x | ...] . [W slices] form of the LSTM gate GEMMs)"""
    segs, ref = [], torch.zeros(M, N, dtype=torch.float64, device=dev)
    for K in Ks:
        A = rnd(*((K, M) if al else (M, K)))
        B = rnd(*((K, N) if bl else (N, K)), scale=0.05)
        segs.append((A, M if al else K, B, N if bl else K, K, 1))
        ref += (A.t() if al else A).double() @ (B if bl else B.t()).double()
    kw = {}
    if epi:
        bias, mask, res = rnd(N), (torch.rand(M, N, generator=g) < 0.9).float().to(dev) / 0.9, rnd(M, N)
        kw = dict(bias=bias, relu=(epi == 'relu'), mul_mask=mask, addend=None if epi == 'relu' else res)
        ref = ref + bias.double()
        if epi == 'relu':
            ref = ref.clamp_min(0) * mask.double()
        else:
            ref = ref * mask.double() + res.double()
    out = torch.empty(M, N, device=dev)
    ops.gemm(segs, M, N, out, a_layout=al, b_layout=bl, ws=ws, **kw)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    K = sum(Ks)
    if time_it and not quick:
        t = timeit(lambda: ops.gemm(segs, M, N, out, a_layout=al, b_layout=bl, ws=ws, **kw))
        print('%-46s %8.1f us %7.1f TF/s  relerr %.1e' % (name, t, 2.0 * M * N * K / t / 1e6, err), flush=True)
    else:
        print('%-46s %s relerr %.1e' % (name, 'ok ' if err < 2e-6 else 'BAD', err), flush=True)
    return err


print('CAPMI_X3_TILE=%s CAPMI_X3W_COST=%s NSW=%s PRIO=%s' % (os.environ.get('CAPMI_X3_TILE', 'auto'), os.environ.get('CAPMI_X3W_COST', 'default'),
                                                        os.environ.get('CAPMI_X3W_NSW', '4'), os.environ.get('CAPMI_X3W_PRIO', '0')))
if short:
    run('txe ffn1   x W^T    [6720x2048] K=512', 6720, 2048, [512], 0, 0)
    run('txe ffn1   +bias+mask+addend', 6720, 2048, [512], 0, 0, epi='add')
    run('txe logit           [6720x9488] K=512', 6720, 9488, [512], 0, 0)
    run('txe dW ffn1 dH^T x  [2048x512]  K=6720', 2048, 512, [6720], 1, 1)
    run('txe dX ffn1 dH W    [6720x512]  K=2048', 6720, 512, [2048], 0, 1)
    run('uxe dW lstm dG^T X  [4000x1000] K=6720', 4000, 1000, [6720], 1, 1)
    run('square              [4096x4096] K=1024', 4096, 4096, [1024], 0, 0)
    run('uxe gates  [h|x|h]  [320x4000]  K=3x1000', 320, 4000, [1000, 1000, 1000], 0, 0)
    run('uxe gates  [h|h]    [320x4000]  K=2x1000', 320, 4000, [1000, 1000], 0, 0)
    run('uxe dX gates dG W   [320x3000]  K=4000', 320, 3000, [4000], 0, 1)
    run('aoa prefill         [360x2048]  K=1024', 360, 2048, [1024], 0, 0)
    sys.exit(0)
worst = 0.0
# ---- edge shapes, every layout (correctness only)
for al in (0, 1):
    for bl in (0, 1):
        for M, N, Ks in ((260, 1156, [36]), (516, 644, [100, 68]), (512, 512, [32]), (772, 520, [40, 32, 36]), (1028, 640, [1000])):
            worst = max(worst, run('edge al=%d bl=%d [%dx%d] K=%s' % (al, bl, M, N, Ks), M, N, Ks, al, bl, time_it=False))
# skinny products (few rows, many columns): planned on the wide kernel with the operands swapped (C^T, x3_epilogue_t) -- 16-byte and
# element-wise epilogue paths (N % 4 != 0), every epilogue operand, both B layouts, several K segments
for bl in (0, 1):
    for M, N, Ks in ((320, 4000, [100, 60]), (260, 1156, [36]), (300, 1032, [64, 32, 32]), (130, 2052, [40])):
        worst = max(worst, run('skinny al=0 bl=%d [%dx%d] K=%s' % (bl, M, N, Ks), M, N, Ks, 0, bl, time_it=False))
worst = max(worst, run('skinny al=0 bl=0 [300x1030] K=[64] (N % 4 != 0)', 300, 1030, [64], 0, 0, time_it=False))
for epi in ('relu', 'add'):
    worst = max(worst, run('skinny epilogue %s [320x1100] K=200' % epi, 320, 1100, [200], 0, 0, epi=epi, time_it=False))
    worst = max(worst, run('skinny epilogue %s [300x1030] K=64 (element-wise)' % epi, 300, 1030, [64], 0, 0, epi=epi, time_it=False))
worst = max(worst, run('edge epilogue relu+mask [772x520] K=200', 772, 520, [200], 0, 0, epi='relu', time_it=False))
worst = max(worst, run('edge epilogue mask+addend [772x520] K=200', 772, 520, [200], 0, 1, epi='add', time_it=False))
assert worst < 2e-6, worst
if quick:
    print('all shapes ok, worst relerr %.1e' % worst)
    sys.exit(0)
# ---- the shapes that carry the XE steps
run('txe ffn1   x W^T    [6720x2048] K=512', 6720, 2048, [512], 0, 0)
run('txe ffn1   +bias+relu+mask', 6720, 2048, [512], 0, 0, epi='relu')
run('txe ffn1   +bias+mask+addend', 6720, 2048, [512], 0, 0, epi='add')
run('txe ffn2   h W^T    [6720x512]  K=2048', 6720, 512, [2048], 0, 0)
run('txe out    x W^T    [6720x512]  K=512', 6720, 512, [512], 0, 0)
run('txe qkv    x W^T    [6720x1536] K=512', 6720, 1536, [512], 0, 0)
run('txe enc ffn1        [2304x2048] K=512', 2304, 2048, [512], 0, 0)
run('txe dW ffn1 dH^T x  [2048x512]  K=6720', 2048, 512, [6720], 1, 1)
run('txe dX ffn1 dH W    [6720x512]  K=2048', 6720, 512, [2048], 0, 1)
run('txe logit           [6720x9488] K=512', 6720, 9488, [512], 0, 0)
run('uxe gates  [h|x|h]  [320x4000]  K=3x1000', 320, 4000, [1000, 1000, 1000], 0, 0)
run('uxe gates  [h|h]    [320x4000]  K=2x1000', 320, 4000, [1000, 1000], 0, 0)
run('uxe xt all steps    [6720x4000] K=1000', 6720, 4000, [1000], 0, 0)
run('uxe logit           [6720x9488] K=1000', 6720, 9488, [1000], 0, 0)
run('uxe dX gates dG W   [320x3000]  K=4000', 320, 3000, [4000], 0, 1)
run('uxe dW lstm dG^T X  [4000x1000] K=6720', 4000, 1000, [6720], 1, 1)
run('uxe dW logit dL^T h [9488x1000] K=6720', 9488, 1000, [6720], 1, 1)
run('scst dW lstm        [4000x1000] K=1000', 4000, 1000, [1000], 1, 1)
run('scst dW logit       [9488x1000] K=1000', 9488, 1000, [1000], 1, 1)
run('scst d_hdrop dL W   [1000x1000] K=9488', 1000, 1000, [9488], 0, 1)
run('square              [4096x4096] K=1024', 4096, 4096, [1024], 0, 0)
run('square              [8192x8192] K=2048', 8192, 8192, [2048], 0, 0)
