#!/usr/bin/env python3
"""Launch the att-LSTM gate GEMM (M rows, 3 K segments of 1000, N=4000) a few times (for rocprofv3 --pmc)."""
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops
dev = torch.device('cuda:0')
R = E = 1000
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ws = ops.Workspace(dev, 32 << 20)
h = torch.randn(M, R, device=dev)
x = torch.randn(M, E, device=dev)
W_ih = torch.randn(4 * R, 2 * R + E, device=dev) * 0.03
W_hh = torch.randn(4 * R, R, device=dev) * 0.03
out = torch.empty(M, 4 * R, device=dev)
segs = [(h, R, W_ih, 2 * R + E, R, 1), (x, E, (W_ih, 2 * R), 2 * R + E, E, 1), (h, R, W_hh, R, R, 1)]
for _ in range(10):
    ops.gemm(segs, M, 4 * R, out, ws=ws, splits=splits, defer_reduce=True)
torch.cuda.synchronize()
