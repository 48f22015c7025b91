#!/usr/bin/env python3
"""Phase timeline of the decode gate GEMM (CAPMI_ARES_ABLATE=16 build variant: s_memtime stamps of waves 0/4 of every
workgroup land in the ticket words of the split-K workspace).  CAPMI_ARES_ABLATE=16 python scripts/gemm_trace.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')
R = E = 1000
M = 60
ws = ops.Workspace(dev, 64 << 20)
h = torch.randn(M, R, device=dev)
x = torch.randn(M, E, device=dev)
Ws = [(torch.randn(4 * R, 2 * R + E, device=dev) * 0.03, torch.randn(4 * R, R, device=dev) * 0.03) for _ in range(3)]
out = torch.empty(M, 4 * R, device=dev)


def run(i):
    W_ih, W_hh = Ws[i % 3]
    segs = [(h, R, W_ih, 2 * R + E, R, 1), (x, E, (W_ih, 2 * R), 2 * R + E, E, 1), (h, R, W_hh, R, R, 1)]
    ops.gemm(segs, M, 4 * R, out, ws=ws, splits=0, defer_reduce=True)


for i in range(12):
    run(i)
torch.cuda.synchronize()
t = ws.buf[:10240].view(torch.int32).cpu().numpy().view(np.uint64).reshape(256, 2, 10).astype(np.float64)
ws.buf[:16384].zero_()
t0 = t[:, :, 0].min()
rel = (t - t0) / 100.0      # s_memtime ticks: 100 MHz constant clock on gfx9 -> 10 ns per tick?  printed raw below too
names = ['entry', 'table+barrier', 'A staged (LDS written)', 'barrier', 'chunk 0 done', 'chunk 2 done', 'loop done', 'K-half reduce', 'slab stored', 'A landed']
# the cycle counters of different XCDs are not synchronised: report every stamp relative to ITS OWN wave's entry stamp
for k in (0, 1):
    print('wave kh=%d: cycles since the wave entered the kernel: median / min / max over 256 workgroups' % k)
    for s_ in (1, 9, 2, 3, 4, 5, 6, 7, 8):
        if k == 1 and s_ == 8:
            continue
        col = t[:, k, s_] - t[:, k, 0]
        print('  %-26s %8.0f %8.0f %8.0f' % (names[s_], np.median(col), col.min(), col.max()))
