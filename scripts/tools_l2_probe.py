#!/usr/bin/env python3
"""Does an XCD's L2 keep weight lines across kernel boundaries?  The loader/consumer decode GEMM (60 rows, K = 3000) on
N columns: the same weight matrix every launch vs three rotating copies (Infinity Cache / HBM)."""
import sys
import ctypes as C
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops
from imagecaptioning.pytorch_amd._lib import lib

dev = torch.device('cuda:0')
M, K = 60, 3000


def timeit(fns, iters=60):
    for f in fns:
        f()
    torch.cuda.synchronize()
    lib.capmi_prof_reset()
    lib.capmi_prof_enable((1 << 0) | (1 << 1) | (1 << 9))
    for i in range(iters):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    lib.capmi_prof_enable(0)
    tot, cnt = 0.0, 0
    for cls in (0, 1, 9):
        ms, n, b_, f_ = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        assert lib.capmi_prof_read(cls, C.byref(ms), C.byref(n), C.byref(b_), C.byref(f_)) == 0
        tot, cnt = tot + ms.value, cnt + n.value
    lib.capmi_prof_reset()
    return tot / max(cnt, 1) * 1e3


ws = ops.Workspace(dev, 64 << 20)
x = torch.randn(M, K, device=dev)
px = ops.planes_from_f32(x)
for N in (512, 1000, 2000, 4000):
    out = torch.empty(M, N, device=dev)
    Ws = [torch.randn(N, K, device=dev) * 0.03 for _ in range(6)]
    mk = lambda W: (lambda: ops.gemm([(x, K, W, K, K, 1)], M, N, out, ws=ws, defer_reduce=True, a_planes=[px]))     # noqa: E731
    same = timeit([mk(Ws[0])])
    rot3 = timeit([mk(W) for W in Ws[:3]])
    rot6 = timeit([mk(W) for W in Ws])
    mb = N * K * 4 / 1e6
    print('N=%4d (%5.1f MB of weights, %.1f MB per XCD): same weights %.1f us | 3 rotating %.1f us | 6 rotating %.1f us'
          % (N, mb, mb / 8, same, rot3, rot6), flush=True)
