#!/usr/bin/env python3
"""Phase ablation of the short-sequence MHA kernels at the Transformer XE (bs64 x 5) shapes: one process per CAPMI_MHA_ABL value
(research build only: CAPMI_LIB=variants/libcapmi.so).  Prints us per launch (forward / backward) per shape; about 12 us of host
time per call is the floor of the forward column.
bits: 1 K/V load, 2 Q load, 4 scores, 8 mask, 16 softmax, 32 p store + dropout load, 64 P V, 128 o store (forward);
256 dP, 512 softmax backward, 1024 dQ, 2048 dK + dV, 4096 P / dropout load (backward).    python scripts/mha_ablate.py [abl ...]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {  # name: (Nq, q_per_kv, Tq, Tk, mask_tq, mask_per_q, causal, fused, D)       (h = 8)
    'enc 64x36x36': (64, 1, 36, 36, 1, 0, 0, True, 512),
    'enc5 320x36x36': (320, 1, 36, 36, 1, 0, 0, True, 512),          # r6: the reference's per-caption encoder (train mode)
    'dec 320x21x21': (320, 1, 21, 21, 21, 1, 1, True, 512),
    'cross 320x21x36': (320, 5, 21, 36, 1, 0, 0, False, 512),
    'aoa_refine 10x36x36': (10, 1, 36, 36, 1, 0, 0, True, 1024),
    'aoa_step 50x1x36': (50, 5, 1, 36, 1, 0, 0, False, 1024),
    'aoa_tf 50x17x36': (50, 5, 17, 36, 1, 0, 0, False, 1024),
}


def child():
    import torch
    from imagecaptioning.pytorch_amd import transformer_engine as TE
    h = 8
    out = []
    for name, (Nq, qpk, Tq, Tk, mtq, mpq, causal, fused, D) in SHAPES.items():
        g = torch.Generator(device='cuda').manual_seed(1)
        Nkv = Nq // qpk
        if fused:
            qkv = torch.randn(Nq * Tq, 3 * D, device='cuda', generator=g)
            q, k, v = (qkv, 0), (qkv, D), (qkv, 2 * D)
            kw = dict(ldkv=Tk * 3 * D, kstride=3 * D, qstride=3 * D)
        else:
            qq = torch.randn(Nq * Tq, D, device='cuda', generator=g)
            kv = torch.randn(Nkv * Tk, 2 * D, device='cuda', generator=g)
            q, k, v = qq, (kv, 0), (kv, D)
            kw = dict(ldkv=Tk * 2 * D, kstride=2 * D, qstride=0)
        mask = torch.ones((Nq if mpq else Nkv) * mtq * Tk, dtype=torch.uint8, device='cuda')
        drop = (torch.rand(Nq * h * Tq * Tk, device='cuda', generator=g) > 0.1).float() / 0.9

        def run():
            return TE.mha_fwd(q, k, v, kw['ldkv'], Nq, qpk, Tq, Tk, h, mask=mask, mask_tq=mtq, mask_per_q=mpq, causal=causal, drop=drop,
                              kstride=kw['kstride'], qstride=kw['qstride'], D=D)
        _, pp = run()
        d_o = torch.randn(Nq, Tq, D, device='cuda', generator=g)
        dq = torch.empty(Nq * Tq, kw['qstride'] or D, device='cuda')
        dkv = torch.empty(Nkv * Tk, 2 * D, device='cuda')

        def run_b():
            TE.mha_bwd(d_o, q, k, v, kw['ldkv'], pp, drop, Nq, qpk, Tq, Tk, h, kstride=kw['kstride'], dk_out=(dkv, 0), dv_out=(dkv, D),
                       dkv_ld=Tk * 2 * D, dkv_stride=2 * D, qstride=kw['qstride'], dq=(dq, 0), dq_stride=kw['qstride'])
        res = []
        for fn in (run, run_b):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
            res.append(best)
        out.append('%s %5.1f /%5.1f' % (name, res[0], res[1]))
    print('abl %5s mfma %2s | ' % (os.environ.get('CAPMI_MHA_ABL', '0'), os.environ.get('CAPMI_MHA_MFMA', '-1')) + ' | '.join(out), flush=True)


if __name__ == '__main__':
    if os.environ.get('MHA_CHILD'):
        child()
    else:
        for a in (sys.argv[1:] or ['0', '1', '2', '4', '8', '16', '32', '64', '128', '255']):
            env = dict(os.environ, MHA_CHILD='1', CAPMI_MHA_ABL=a)
            subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
