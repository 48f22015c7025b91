"""Grouped weight-gradient GEMMs (capmi_gemm_group_tn) against the same GEMMs launched one by one (ops.DeferredGrads' r5 form: a
deferred-reduction GEMM each + ONE batched reduction), same process, HIP-event timed.
    python scripts/tools_group_bench.py [scst|txe|txe_faithful|aoa|uxe] ..."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioning.pytorch_amd import ops  # noqa: E402

SETS = {
    'scst': [(4000, 1000, 1000)] * 6 + [(512, 1000, 1000), (9488, 1000, 1000)],
    'uxe': [(4000, 1000, 6720)] * 6 + [(512, 1000, 6720), (9488, 1000, 6720)],
    'txe': ([(512, 512, 2304)] * 4 + [(2048, 512, 2304), (512, 2048, 2304)]) * 6 + ([(512, 512, 6720)] * 8 + [(2048, 512, 6720), (512, 2048, 6720)]) * 6,
    'txe_faithful': ([(512, 512, 11520)] * 4 + [(2048, 512, 11520), (512, 2048, 11520)]) * 6 + ([(512, 512, 6720)] * 8 + [(2048, 512, 6720), (512, 2048, 6720)]) * 6,
    'aoa': ([(1024, 1024, 360)] * 4 + [(2048, 2048, 360)]) * 6 + [(4096, 1024, 1000), (4096, 2048, 1000), (2048, 2048, 1000)],
}


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device('cuda:0')
    names = sys.argv[1:] or sorted(SETS)
    for name in names:
        if name not in SETS and name.count('x') == 3:                     # MxNxKxcount: `count` identical GEMMs (unit-length sweeps: time = a + rounds * (K / 32 * s + c))
            M, N, K, cnt = (int(v) for v in name.split('x'))
            SETS[name] = [(M, N, K)] * cnt
        shapes = SETS[name]
        g = torch.Generator().manual_seed(1)
        items = [(torch.randn(K, M, generator=g).to(dev), torch.randn(K, N, generator=g).to(dev), torch.empty(M, N, device=dev), False)
                 for M, N, K in shapes]
        flops = sum(2.0 * M * N * K for M, N, K in shapes)

        def grouped():
            ops.gemm_group_tn(items, cache_key=('bench', name))

        def separate():
            for dy, x, out, _ in items:
                K, M = dy.shape
                ops.gemm([(dy, M, x, x.shape[1], K, 1)], M, x.shape[1], out, a_layout=1, b_layout=1)

        def deferred():
            os.environ['CAPMI_DW_GROUP'] = '0'
            os.environ['CAPMI_DW_STREAM'] = '0'
            d = ops.DeferredGrads(dev)
            for dy, x, out, _ in items:
                d.dw(dy, x, out)
            d.flush()

        tg, ts, td = timed(grouped), timed(separate), timed(deferred)
        print('%-13s %3d GEMMs %7.1f GFLOP | grouped %8.1f us %6.1f TF | one by one %8.1f us %6.1f TF | deferred (r5) %8.1f us %6.1f TF'
              % (name, len(shapes), flops / 1e9, tg, flops / tg / 1e6, ts, flops / ts / 1e6, td, flops / td / 1e6), flush=True)


if __name__ == '__main__':
    main()
