#!/usr/bin/env python3
"""Roofline of the fused region-attention kernel at eval/XE batch sizes (HIP events around batches of launches)."""
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')
K, A, R = 36, 512, 1000


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


for B, n in [(10, 6), (64, 5), (256, 1), (1024, 1), (2048, 1), (4096, 1), (1024, 5)]:
    N = B * n
    att_h = torch.randn(N, A, device=dev)
    p_att = torch.randn(B, K, A, device=dev)
    att = torch.randn(B, K, R, device=dev)
    w = torch.randn(A, device=dev) * 0.1
    b = torch.zeros(1, device=dev)
    us = timeit(lambda: ops.attention_fwd(att_h, p_att, att, None, w, b, n))
    byts = 4.0 * (B * K * (A + R) + N * (A + R + K))
    print('B=%5d n=%d: %8.1f us  unique %.1f MB  -> %.2f TB/s (%.1f%% of 8 TB/s)' % (B, n, us, byts / 1e6, byts / us / 1e6, byts / us / 1e6 / 8 * 100),
          flush=True)
