import os, sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import synthetic, _lib
from imagecaptioning.pytorch_amd.captioning import models
from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper
from imagecaptioning.pytorch_amd.captioning.utils import rewards
dev = torch.device('cuda:0')
opt = synthetic.updown_opt()
torch.manual_seed(1234)
model = models.setup(opt).to(dev)
flat = model.flatten_parameters_()
lw = LossWrapper(model, opt)
B = 10
fc, att = synthetic.batch(B, seed=1234, device=dev)
corpus = synthetic.corpus(2000, seed=7)
df, ref_len = synthetic.document_frequency(corpus)
rewards.reset_scorer(); rewards.init_scorer((df, ref_len), device=dev)
gts = synthetic.corpus(B, seed=100)
gi = torch.arange(B)
for it in range(16):
    out = lw(fc, att, None, None, None, gts, gi, True, False, False)
    torch.cuda.synchronize(); print(it, 'fwd ok', float(out['loss']), flush=True)
    loss = out['loss'].mean()
    flat.zero_grad()
    loss.backward()
    torch.cuda.synchronize(); print(it, 'bwd ok', flush=True)
    flat.collect_grads()
    bad = [n for n in flat.names if not torch.isfinite(flat.grad_views[n]).all()]
    print(it, 'grad norm', float(flat.grad.norm()), 'nonfinite grads:', bad, flush=True)
    flat.adam_step(5e-4, clip_value=0.1)
    torch.cuda.synchronize(); print(it, 'adam ok', 'params finite', bool(torch.isfinite(flat.flat).all()), 'logp finite', bool(torch.isfinite(model._last_rollout.seq_logp).all()), 'tok max', int(model._last_rollout.seq.max()), flush=True)
