#!/usr/bin/env python3
"""Micro-benchmark of the decode-step GEMM shapes (HIP events around batches of launches)."""
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')
R = E = 1000
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def timeit(fn, iters=50):
    for _ in range(5):
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
h = torch.randn(M, R, device=dev)
x = torch.randn(M, E, device=dev)
W_ih = torch.randn(4 * R, 2 * R + E, device=dev) * 0.03
W_hh = torch.randn(4 * R, R, device=dev) * 0.03
Wl = torch.randn(9488, R, device=dev) * 0.03
out = torch.empty(M, 4 * R, device=dev)
outl = torch.empty(M, 9488, device=dev)
dg = torch.randn(M, 4 * R, device=dev)
dx = torch.empty(M, 2 * R, device=dev)
segs = [(h, R, W_ih, 2 * R + E, R, 1), (x, E, (W_ih, 2 * R), 2 * R + E, E, 1), (h, R, W_hh, R, R, 1)]
Wlang = torch.randn(4 * R, 2 * R, device=dev) * 0.03
Wh = torch.randn(512, R, device=dev) * 0.03
Wcat3 = torch.randn(4 * R, 3 * R, device=dev) * 0.03
dx3 = torch.empty(M, 3 * R, device=dev)
outh = torch.empty(M, 512, device=dev)
for splits in [int(s) for s in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0'])]:
    t1 = timeit(lambda: ops.gemm(segs, M, 4 * R, out, ws=ws, splits=splits, defer_reduce=True))
    t2 = timeit(lambda: ops.gemm([(h, R, Wl, R, R, 1)], M, 9488, outl, ws=ws, splits=max(1, splits // 3) if splits else 0))
    t3 = timeit(lambda: ops.gemm([(dg, 4 * R, Wlang, 2 * R, 4 * R, 1)], M, 2 * R, dx, a_layout=0, b_layout=1, ws=ws, splits=splits * 2))
    t4 = timeit(lambda: ops.gemm([(h, R, Wh, R, R, 1)], M, 512, outh, ws=ws, splits=0))
    t5 = timeit(lambda: ops.gemm([(dg, 4 * R, Wcat3, 3 * R, 4 * R, 1)], M, 3 * R, dx3, a_layout=0, b_layout=1, ws=ws, defer_reduce=True))
    print('M=%d splits=%d: gates(deferred) %.1f us (%.2f TB/s)  logit %.1f us  dX_nn %.1f us  h2att %.1f us  dX2_nn(deferred) %.1f us' %
          (M, splits, t1, 48e6 / t1 / 1e6, t2, t3, t4, t5), flush=True)
