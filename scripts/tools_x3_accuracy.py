import sys, torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import ops
dev = torch.device('cuda:0')
ws = ops.Workspace(dev, 64 << 20)
torch.manual_seed(0)
for (M, N, K, al, bl) in [(4000, 1000, 1200, 1, 1), (1200, 1000, 9488, 0, 1), (512, 768, 4096, 0, 0)]:
    A = torch.randn((K, M) if al else (M, K), device=dev)
    B = torch.randn((K, N) if bl else (N, K), device=dev)
    # wide dynamic range too
    A = A * torch.exp(torch.randn_like(A) * 2)
    out = torch.empty(M, N, device=dev)
    ops.gemm([(A, M if al else K, B, N if bl else K, K, 1)], M, N, out, a_layout=al, b_layout=bl, ws=ws)
    A64 = (A.t() if al else A).double(); B64 = (B if bl else B.t()).double()
    ref = A64 @ B64
    mag = (A64.abs() @ B64.abs())
    t32 = ((A.t() if al else A) @ (B if bl else B.t())).double()
    e_x3 = ((out.double() - ref).abs() / mag).max().item()
    e_t32 = ((t32 - ref).abs() / mag).max().item()
    print(M, N, K, 'x3 err/|a||b| %.2e   torch fp32 %.2e' % (e_x3, e_t32))
