import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_graph_step_gpu as T
from imagecaptioning.pytorch_amd.graph_step import TrainStep
fam = sys.argv[1] if len(sys.argv) > 1 else 'transformer'
flags = (False, False) if fam == 'transformer' else (False, True)
runs = {}
for mode in ('stepped', 'graph'):
    opt, model, flat, lw, dims = T._setup(fam)
    batches = T._batches(fam, dims)
    ts = TrainStep(lw, flat, opt, T.DEV, graph=(mode == 'graph'))
    snaps = []
    for it in range(4):
        loss, out = ts(batches[it % 3], flags[0], flags[1], lr=1e-3)
        snaps.append((loss.clone().cpu(), flat.grad.clone().cpu(), flat.flat.clone().cpu()))
    runs[mode] = (snaps, flat)
    print(mode, ts.captures, ts.replays, ts.stepped, ts.failed)
a, fa = runs['stepped']; b, fb = runs['graph']
for it in range(4):
    print('step', it, 'loss eq', torch.equal(a[it][0], b[it][0]), 'grad eq', torch.equal(a[it][1], b[it][1]), 'param eq', torch.equal(a[it][2], b[it][2]))
    if not torch.equal(a[it][1], b[it][1]):
        for n, p, o in zip(fa.names, fa.params, fa.offsets):
            ga, gb = a[it][1][o:o + p.numel()], b[it][1][o:o + p.numel()]
            if not torch.equal(ga, gb):
                print('   grad', n, float((ga - gb).abs().max()), float(ga.abs().max()))
