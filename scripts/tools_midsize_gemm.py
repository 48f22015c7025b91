#!/usr/bin/env python3
"""Mid-size GEMMs of the AoA refiner / prefill (360 region rows = 10 images x 36): time per launch under the planner's choice.
    python scripts/tools_midsize_gemm.py            # default planner
    CAPMI_GEMM_X3=0 python scripts/tools_midsize_gemm.py     # exact-fp32 MFMA tiles instead of the bf16x3 fat kernel
One line per shape: us per call of ops.gemm (reduce launch included), GFLOP, error against fp64."""
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


def run(name, M, N, K, al, bl):
    A = torch.randn(*((K, M) if al else (M, K)), generator=g).to(dev)
    B = (torch.randn(*((K, N) if bl else (N, K)), generator=g) * 0.05).to(dev)
    out = torch.empty(M, N, device=dev)
    seg = [(A, M if al else K, B, N if bl else K, K, 1)]
    f = lambda: ops.gemm(seg, M, N, out, a_layout=al, b_layout=bl)      # noqa: E731
    f()
    ref = (A.t() if al else A).double() @ (B if bl else B.t()).double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    t = timeit(f)
    print('%-44s %7.1f us  %6.2f GF  %6.1f TF/s  relerr %.1e' % (name, t, 2e-9 * M * N * K, 2e-6 * M * N * K / t, err), flush=True)


for M in (360, 1000):
    run('fwd  x W^T   [%dx1024] K=1024' % M, M, 1024, 1024, 0, 0)
    run('fwd  x W^T   [%dx2048] K=1024' % M, M, 2048, 1024, 0, 0)
    run('fwd  x W^T   [%dx3072] K=1024' % M, M, 3072, 1024, 0, 0)
    run('fwd  x W^T   [%dx2048] K=2048' % M, M, 2048, 2048, 0, 0)
    run('dX   dy W    [%dx1024] K=3072' % M, M, 1024, 3072, 0, 1)
    run('dX   dy W    [%dx2048] K=2048' % M, M, 2048, 2048, 0, 1)
    run('dW   dy^T x  [2048x2048] K=%d' % M, 2048, 2048, M, 1, 1)
    run('dW   dy^T x  [3072x1024] K=%d' % M, 3072, 1024, M, 1, 1)
    run('dW   dy^T x  [1024x1024] K=%d' % M, 1024, 1024, M, 1, 1)
