import os, sys, torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops
dev = torch.device('cuda:0')
M, N, K, cnt = 4096, 1024, 512, 8     # 8 x 128 tiles = 1024 units = 4 rounds of 16 K tiles
g = torch.Generator().manual_seed(1)
items = [(torch.randn(K, M, generator=g).to(dev), torch.randn(K, N, generator=g).to(dev), torch.empty(M, N, device=dev), False) for _ in range(cnt)]
for _ in range(3):
    ops.gemm_group_tn(items)
torch.cuda.synchronize()
