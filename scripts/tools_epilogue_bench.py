#!/usr/bin/env python3
"""What the epilogue operands (bias, ReLU, dropout mask, residual addend) cost a fat GEMM at the Transformer XE shapes."""
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')


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
for name, M, N, K in [('ffn1 [6720x2048] K=512', 6720, 2048, 512), ('ffn2 [6720x512] K=2048', 6720, 512, 2048),
                      ('out  [6720x512] K=512', 6720, 512, 512), ('enc ffn1 [2304x2048] K=512', 2304, 2048, 512)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    mask = (torch.rand(M, N, device=dev) < 0.9).float() / 0.9
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    seg = [(x, K, w, K, K, 1)]
    t0 = timeit(lambda: ops.gemm(seg, M, N, out, ws=ws))
    t1 = timeit(lambda: ops.gemm(seg, M, N, out, ws=ws, bias=b, relu=True))
    t2 = timeit(lambda: ops.gemm(seg, M, N, out, ws=ws, bias=b, relu=True, mul_mask=mask))
    t3 = timeit(lambda: ops.gemm(seg, M, N, out, ws=ws, bias=b, mul_mask=mask, addend=res))
    print('%-28s plain %6.1f | bias+relu %6.1f | +mask %6.1f | bias+mask+addend %6.1f us   (%.0f TF plain)' %
          (name, t0, t1, t2, t3, 2.0 * M * N * K / t0 / 1e6), flush=True)
