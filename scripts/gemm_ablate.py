#!/usr/bin/env python3
"""Ablation timing of the decode-step gate GEMM (48 MB, M=60): run once per CAPMI_ARES_ABLATE value (read at library load).
    for a in 0 1 2 3 4 6 7 8 10 15; do CAPMI_ARES_ABLATE=$a python scripts/gemm_ablate.py; done"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')
R = E = 1000
M = 60
ws = ops.Workspace(dev, 64 << 20)
h = torch.randn(M, R, device=dev)
x = torch.randn(M, E, device=dev)
# three weight sets rotate like a decode step does (att gates, lang gates, logit: 136 MB -- Infinity-Cache resident)
Ws = [(torch.randn(4 * R, 2 * R + E, device=dev) * 0.03, torch.randn(4 * R, R, device=dev) * 0.03) for _ in range(3)]
out = torch.empty(M, 4 * R, device=dev)


def run(i):
    W_ih, W_hh = Ws[i % 3]
    segs = [(h, R, W_ih, 2 * R + E, R, 1), (x, E, (W_ih, 2 * R), 2 * R + E, E, 1), (h, R, W_hh, R, R, 1)]
    ops.gemm(segs, M, 4 * R, out, ws=ws, splits=0, defer_reduce=True)


for i in range(9):
    run(i)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 90
a.record()
for i in range(iters):
    run(i)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / iters * 1e3
print('CAPMI_ARES_ABLATE=%-3s gate GEMM %.2f us  (%.2f TB/s of 48 MB)' % (os.environ.get('CAPMI_ARES_ABLATE', '0'), us, 48.0 / us), flush=True)
