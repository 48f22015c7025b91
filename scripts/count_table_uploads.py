import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from imagecaptioning.pytorch_amd import ops
from imagecaptioning.pytorch_amd.captioning import models
from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper
from imagecaptioning.pytorch_amd import synthetic
dev = torch.device('cuda:0')
opt = bench._opt('transformer'); opt.vocab = {str(i): 'w%d' % i for i in range(1, synthetic.VOCAB + 1)}
torch.manual_seed(1)
model = models.setup(opt).to(dev); flat = model.flatten_parameters_(); model.train()
lw = LossWrapper(model, opt)
B = 64
fc, att = synthetic.batch(B, seed=1, device=dev)
lab, msk = synthetic.xe_labels(B, n=5, L=20, seed=2); lab, msk = lab.to(dev), msk.to(dev)
prev = 0
for it in range(14):
    t0 = time.perf_counter()
    out = lw(fc, att, lab, msk, None, None, torch.arange(B), False, False, False)
    flat.zero_grad(); out['loss'].mean().backward(); flat.collect_grads()
    flat.adam_step(lr=1e-4, clip_value=0.1)
    t1 = time.perf_counter()
    print('step', it, 'table uploads', ops.DeferredGrads.uploads - prev, 'host ms %.1f' % ((t1 - t0) * 1e3), flush=True)
    prev = ops.DeferredGrads.uploads
torch.cuda.synchronize()
