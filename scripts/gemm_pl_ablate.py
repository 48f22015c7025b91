#!/usr/bin/env python3
"""Ablation timing / phase trace of the A-planes gate GEMM (48 MB, M=60), one process per CAPMI_APL_ABLATE / CAPMI_APL_PF
setting (read at library load):  CAPMI_APL_ABLATE=3 python scripts/gemm_pl_ablate.py ;  CAPMI_APL_ABLATE=16 ... prints the trace."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioning.pytorch_amd import ops
from imagecaptioning.pytorch_amd._lib import lib

dev = torch.device('cuda:0')
R = E = 1000
M = 60
abl = int(os.environ.get('CAPMI_APL_ABLATE', '0')) | int(os.environ.get('CAPMI_LC_ABLATE', '0'))
lc = os.environ.get('CAPMI_LC', '1') != '0'
ws = ops.Workspace(dev, 64 << 20)
h, x = torch.randn(M, R, device=dev), torch.randn(M, E, device=dev)
ph, px = ops.planes_from_f32(h), ops.planes_from_f32(x)
Ws = [(torch.randn(4 * R, 2 * R + E, device=dev) * 0.03, torch.randn(4 * R, R, device=dev) * 0.03) for _ in range(3)]
out = torch.empty(M, 4 * R, device=dev)


def run(i):
    W_ih, W_hh = Ws[i % 3]
    segs = [(h, R, W_ih, 2 * R + E, R, 1), (x, E, (W_ih, 2 * R), 2 * R + E, E, 1), (h, R, W_hh, R, R, 1)]
    ops.gemm(segs, M, 4 * R, out, ws=ws, splits=0, defer_reduce=True, a_planes=[ph, px, ph])


for i in range(12):
    run(i)
torch.cuda.synchronize()
if abl & 16:
    t = ws.buf[:10240].view(torch.int32).cpu().numpy().view(np.uint64).reshape(256, 2, 10).astype(np.float64)
    ws.buf[:16384].zero_()
    if lc:
        for k, names, order in ((0, ['entry', 'consumer set up', 'loop done', 'slab stored', 'stage 0 handed over'], (1, 4, 2, 3)),
                                (1, ['entry', 'first 2 stages requested', 'stage 0 landed', 'all barriers passed'], (1, 2, 3))):
            print('loader / consumer kernel, %s wave: cycles since the wave entered the kernel: median / min / max over 256 workgroups'
                  % ('loader' if k else 'consumer'))
            for s_ in order:
                col = t[:, k, s_] - t[:, k, 0]
                print('  %-30s %8.0f %8.0f %8.0f' % (names[s_], np.median(col), col.min(), col.max()))
        sys.exit(0)
    names = ['entry', 'DMAs issued', 'weights requested', 'A landed (vmcnt)', 'barrier passed', 'chunk 0 done', 'loop done',
             'K-half reduce', 'slab stored']
    for k in (0, 1):
        print('PF=%s wave kh=%d: cycles since the wave entered the kernel: median / min / max over 256 workgroups'
              % (os.environ.get('CAPMI_APL_PF', '0'), k))
        for s_ in range(1, 9):
            if k == 1 and s_ == 8:
                continue
            col = t[:, k, s_] - t[:, k, 0]
            print('  %-22s %8.0f %8.0f %8.0f' % (names[s_], np.median(col), col.min(), col.max()))
    sys.exit(0)
lib.capmi_prof_reset()
lib.capmi_prof_enable(1 << 9)
iters = 90
for i in range(iters):
    run(i)
torch.cuda.synchronize()
lib.capmi_prof_enable(0)
ms, n, b_, f_ = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
assert lib.capmi_prof_read(9, C.byref(ms), C.byref(n), C.byref(b_), C.byref(f_)) == 0 and n.value == iters
us = ms.value / iters * 1e3
print('%s ABLATE=%-3s PF=%s gate GEMM %.2f us  (%.2f TB/s of 48.6 MB)'
      % ('LC ' if lc else 'APL', abl, os.environ.get('CAPMI_APL_PF', '0'), us, 48.6 / us), flush=True)
