"""Where do the __amd_rocclr_copyBuffer launches of a training step come from?  Runs a few steps of a bench configuration under
torch.profiler (with stacks) and lists every device copy / ATen kernel with its size, duration and the Python frames that issued it.
    python scripts/find_copies.py [updown_scst|aoa_nsc|transformer_xe]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('CAPMI_GRAPH_STEP', '0')           # stepped: the profiler sees the Python frames
import bench  # noqa: E402
from imagecaptioning.pytorch_amd import synthetic  # noqa: E402
from imagecaptioning.pytorch_amd.captioning import models  # noqa: E402
from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper  # noqa: E402
from imagecaptioning.pytorch_amd.captioning.utils import rewards  # noqa: E402
from imagecaptioning.pytorch_amd.graph_step import TrainStep  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'updown_scst'
    cfg = bench.CONFIGS[name]
    dev = torch.device('cuda:0')
    opt = bench._opt(cfg[0])
    sc, st = cfg[2]
    torch.manual_seed(1234)
    model = models.setup(opt).to(dev)
    flat = model.flatten_parameters_()
    lw = LossWrapper(model, opt)
    B, L = cfg[1], opt.max_length
    rewards.reset_scorer()
    if sc or st:
        rewards.init_scorer(synthetic.document_frequency(synthetic.corpus(2000, seed=7)), device=dev)
    f_, a_ = synthetic.batch(B, seed=1, device=dev)
    lab = msk = None
    if not (sc or st):
        lab, msk = synthetic.xe_labels(B, n=5, L=L, seed=1)
        lab, msk = lab.to(dev), msk.to(dev)
    data = {'fc_feats': f_, 'att_feats': a_, 'att_masks': None, 'labels': lab, 'masks': msk, 'gts': rewards.pack_gts(synthetic.corpus(B, seed=3))}
    ts = TrainStep(lw, flat, opt, dev, graph=False)
    for _ in range(4):
        ts(data, sc, st)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(2):
            ts(data, sc, st)
        torch.cuda.synchronize()
    rows = []
    for ev in prof.events():
        n = ev.name
        if ('copy' in n.lower() or 'Memcpy' in n or 'Memset' in n or n.startswith('aten::')) and ev.device_time_total > 0 and not ev.cpu_children:
            stack = [f for f in (ev.stack or []) if 'imagecaptioning' in f or 'bench.py' in f][:3]
            rows.append((ev.device_time_total, n, str(ev.input_shapes)[:80], ' <- '.join(s.strip()[-70:] for s in stack)))
    rows.sort(reverse=True)
    print('device us (2 steps) | op | shapes | frames')
    for r in rows[:40]:
        print('%9.1f | %s | %s | %s' % r)


if __name__ == '__main__':
    main()
