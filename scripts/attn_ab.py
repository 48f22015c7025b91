#!/usr/bin/env python3
"""Fused region-attention forward at the SCST benchmark shape (fused rollout: 60 rows, 20 feature tiles, h2att slabs) and at
XE / evaluation batches.  Run twice: CAPMI_ATT_V2=0 / 1."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioning.pytorch_amd import ops
from imagecaptioning.pytorch_amd._lib import lib, ptr, check, stream_ptr

dev = torch.device('cuda:0')
K, A, R = 36, 512, 1000


import ctypes as C


def timeit(fn, iters=100):
    """mean kernel duration from the in-dispatch HIP events (class 3 = attention forward): no host launch overhead inside"""
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    lib.capmi_prof_reset()
    lib.capmi_prof_enable(1 << 3)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    lib.capmi_prof_enable(0)
    ms, n, b, f = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    assert lib.capmi_prof_read(3, C.byref(ms), C.byref(n), C.byref(b), C.byref(f)) == 0 and n.value == iters
    return ms.value / iters * 1e3


w = torch.randn(A, device=dev) * 0.1
bb = torch.zeros(1, device=dev)
# SCST fused rollout: 50 sampled rows on tiles 0..9, 10 greedy rows on tiles 10..19, att_h as 16 h2att slabs
B, n = 10, 5
N = B * n + B
row_img = torch.cat([torch.arange(B * n) // n, B + torch.arange(B)]).to(torch.int32).to(dev)
p_att = torch.randn(2 * B, K, A, device=dev)
att = torch.randn(2 * B, K, R, device=dev)
for splits in (16, 8):
    slabs = torch.randn(splits, N, A, device=dev)
    hb = torch.randn(A, device=dev)
    att_h = torch.empty(N, A, device=dev)
    ctx, alpha = torch.empty(N, R, device=dev), torch.empty(N, K, device=dev)
    fn = lambda: check(lib.capmi_attention_fwd_partial(ptr(slabs), splits, N * A, ptr(hb), ptr(att_h), ptr(p_att), ptr(att), None,  # noqa: E731
                                                       ptr(w), ptr(bb), ptr(ctx), ptr(alpha), 2 * B, n, K, A, R, ptr(row_img), N,
                                                       stream_ptr()), 'att')
    us = timeit(fn)
    uniq = 4.0 * (2 * B * K * (A + R) + N * (A + R + K) + splits * N * A)
    print('V2=%s SCST fused (60 rows, %2d slabs): %6.2f us  %.2f TB/s of unique bytes' % (os.environ.get('CAPMI_ATT_V2', '1'), splits, us, uniq / us / 1e6))
for (B, n) in ((64, 5), (256, 1), (512, 1), (1024, 1)):
    att_h = torch.randn(B * n, A, device=dev)
    p_att = torch.randn(B, K, A, device=dev)
    att = torch.randn(B, K, R, device=dev)
    us = timeit(lambda: ops.attention_fwd(att_h, p_att, att, None, w, bb, n), 50)
    uniq = 4.0 * (B * K * (A + R) + B * n * (A + R + K))
    print('V2=%s B=%4d n=%d: %6.2f us  %.2f TB/s of unique bytes' % (os.environ.get('CAPMI_ATT_V2', '1'), B, n, us, uniq / us / 1e6))
