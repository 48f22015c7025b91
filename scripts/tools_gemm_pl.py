#!/usr/bin/env python3
"""Decode-step GEMM shapes with the activations delivered as A planes (LDS-DMA staging, round 3) vs the in-kernel split.
Rotates over several weight copies so that a launch streams from the Infinity Cache / HBM like in the step, not from L2."""
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops

dev = torch.device('cuda:0')
R = E = 1000
M = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NW = 3          # rotating weight copies (3 x 48 MB: what a decode step cycles through)


import ctypes as C
from imagecaptioning.pytorch_amd._lib import lib


def timeit(fns, iters=60):
    """average KERNEL duration from the in-dispatch HIP events of the library (classes 0 / 1 / 9: decode + BPTT GEMMs)"""
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
    assert cnt == iters, (cnt, iters)
    return tot / iters * 1e3


ws = ops.Workspace(dev, 64 << 20)
h, x, dg = torch.randn(M, R, device=dev), torch.randn(M, E, device=dev), torch.randn(M, 4 * R, device=dev)
ph, px, pdg = ops.planes_from_f32(h), ops.planes_from_f32(x), ops.planes_from_f32(dg)
out, outl, dx3, dx2, outh = (torch.empty(M, n, device=dev) for n in (4 * R, 9488, 3 * R, 2 * R, 512))
W_ih = [torch.randn(4 * R, 2 * R + E, device=dev) * 0.03 for _ in range(NW)]
W_hh = [torch.randn(4 * R, R, device=dev) * 0.03 for _ in range(NW)]
Wl = [torch.randn(9488, R, device=dev) * 0.03 for _ in range(NW)]
W3 = [torch.randn(4 * R, 3 * R, device=dev) * 0.03 for _ in range(NW)]
W2 = [torch.randn(4 * R, 2 * R, device=dev) * 0.03 for _ in range(NW)]
Wh = torch.randn(512, R, device=dev) * 0.03
for pl in (False, True):
    def gates(i):
        segs = [(h, R, W_ih[i], 2 * R + E, R, 1), (x, E, (W_ih[i], 2 * R), 2 * R + E, E, 1), (h, R, W_hh[i], R, R, 1)]
        return lambda: ops.gemm(segs, M, 4 * R, out, ws=ws, defer_reduce=True, a_planes=[ph, px, ph] if pl else None)
    t1 = timeit([gates(i) for i in range(NW)])
    t2 = timeit([(lambda i=i: ops.gemm([(h, R, Wl[i], R, R, 1)], M, 9488, outl, ws=ws, defer_reduce=True,
                                       a_planes=[ph] if pl else None)) for i in range(NW)])
    t3 = timeit([(lambda i=i: ops.gemm([(dg, 4 * R, W3[i], 3 * R, 4 * R, 1)], M, 3 * R, dx3, b_layout=1, ws=ws, defer_reduce=True,
                                       a_planes=[pdg] if pl else None)) for i in range(NW)])
    t4 = timeit([(lambda i=i: ops.gemm([(dg, 4 * R, W2[i], 2 * R, 4 * R, 1)], M, 2 * R, dx2, b_layout=1, ws=ws, defer_reduce=True,
                                       a_planes=[pdg] if pl else None)) for i in range(NW)])
    t5 = timeit([lambda: ops.gemm([(h, R, Wh, R, R, 1)], M, 512, outh, ws=ws, defer_reduce=True, a_planes=[ph] if pl else None)])
    print('M=%d planes=%d: gates 48.6MB %.1f us (%.2f TB/s) | logit 40.5MB %.1f us (%.2f TB/s) | dX3 48MB %.1f us | dX2 32MB %.1f us | '
          'h2att %.1f us' % (M, pl, t1, 48.6 / t1, t2, 40.5 / t2, t3, t4, t5), flush=True)
