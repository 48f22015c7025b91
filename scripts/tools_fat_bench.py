#!/usr/bin/env python3
"""Micro-benchmark of the time-batched BPTT GEMM shapes (weight gradients and batched dX)."""
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
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


ws = ops.Workspace(dev, 64 << 20)
TN, R, V1 = 1000, 1000, 9488
# (name, M, N, K, a_layout, b_layout)
shapes = [('dW_lstm  dG^T X   [4000x1000] K=1200', 4 * R, R, TN, 1, 1),
          ('dW_logit dL^T h   [9488x1000] K=1200', V1, R, TN, 1, 1),
          ('d_hdrop  dL W     [1200x1000] K=9488', TN, R, V1, 0, 1),
          ('d_xt     dG W     [1200x1000] K=4000', TN, R, 4 * R, 0, 1),
          ('square   A B^T    [4096x4096] K=1024', 4096, 4096, 1024, 0, 0),
          ('square   A B^T    [4096x4096] K=1056', 4096, 4096, 1056, 0, 0),
          ('ffn1     x W^T    [5440x2048] K=512 ', 5440, 2048, 512, 0, 0),
          ('ffn1     x W^T    [5440x2048] K=528 ', 5440, 2048, 528, 0, 0),
          ('ffn2     h W^T    [5440x512 ] K=2048', 5440, 512, 2048, 0, 0),
          ('ffn2     h W^T    [5440x512 ] K=2080', 5440, 512, 2080, 0, 0),
          ('dW_ffn1  dH^T x   [2048x512 ] K=5440', 2048, 512, 5440, 1, 1),
          ('dx_ffn1  dH W     [5440x512 ] K=2048', 5440, 512, 2048, 0, 1)]
for name, M, N, K, al, bl in shapes:
    A = torch.randn((K, M) if al else (M, K), device=dev)
    B = torch.randn((K, N) if bl else (N, K), device=dev)
    out = torch.empty(M, N, device=dev)
    lda = M if al else K
    ldb = N if bl else K
    t = timeit(lambda: ops.gemm([(A, lda, B, ldb, K, 1)], M, N, out, a_layout=al, b_layout=bl, ws=ws))
    ref = (A.t() if al else A) @ (B if bl else B.t())
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    print('%-40s %7.1f us  %6.1f TF/s  relerr %.1e' % (name, t, 2.0 * M * N * K / t / 1e6, err), flush=True)
