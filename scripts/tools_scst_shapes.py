import sys, torch, time
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import synthetic
from imagecaptioning.pytorch_amd.captioning import models
from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper
from imagecaptioning.pytorch_amd.captioning.utils import rewards
dev = torch.device('cuda:0')
for n, B in ((16, 10), (5, 64), (1, 3)):
    opt = synthetic.updown_opt(train_sample_n=n)
    torch.manual_seed(1)
    model = models.setup(opt).to(dev); flat = model.flatten_parameters_(); lw = LossWrapper(model, opt)
    fc, att = synthetic.batch(B, seed=3, device=dev)
    corpus = synthetic.corpus(500, seed=7); df, ref_len = synthetic.document_frequency(corpus)
    rewards.reset_scorer(); rewards.init_scorer((df, ref_len), device=dev)
    gts = synthetic.corpus(B, seed=100)
    for it in range(3):
        out = lw(fc, att, None, None, None, gts, torch.arange(B), True, False, False)
        loss = out['loss'].mean(); flat.zero_grad(); loss.backward(); flat.collect_grads(); flat.adam_step(5e-4, clip_value=0.1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(5):
        out = lw(fc, att, None, None, None, gts, torch.arange(B), True, False, False)
        loss = out['loss'].mean(); flat.zero_grad(); loss.backward(); flat.collect_grads(); flat.adam_step(5e-4, clip_value=0.1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    g = flat.grad
    print('B=%d n=%d: loss %.4f reward %.4f  %.2f ms/step  %.0f captions/s  grad finite %s' % (B, n, float(loss.detach()), float(out['reward'].mean()), dt * 1e3, B * n / dt, bool(torch.isfinite(g).all())), flush=True)
