#!/usr/bin/env python3
"""One launch of the plane-fed fat GEMM with the in-kernel clock probe (CAPMI_GEMM_ABLATE=4: full kernel, 6: without DMAs)."""
import sys
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops
dev = torch.device('cuda:0')
M, N, K = 4000, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
apl, bpl = ops.planes_split(A, transposed=True), ops.planes_split(B, transposed=True)
out = torch.empty(M, N, device=dev)
for _ in range(4):
    ops.gemm([(None, M, None, N, K, 1)], M, N, out, a_layout=1, b_layout=1, a_planes=[apl], b_planes=[bpl])
    torch.cuda.synchronize()
