#!/usr/bin/env python3
"""BASELINE configs[4]: AoA new-self-critical step (bs10 x train_sample_n 5, structure_loss_type new_self_critical,
configs/aoa_nsc.yml) and beam-5 evaluation at the stated model size; secondary workload, see DESIGN.md section 6."""
import sys
import time

import torch

sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import synthetic                                   # noqa: E402
from imagecaptioning.pytorch_amd.captioning import models                          # noqa: E402
from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper  # noqa: E402
from imagecaptioning.pytorch_amd.captioning.utils import rewards                  # noqa: E402

dev = torch.device('cuda:0')
B, n = 10, 5
opt = synthetic.updown_opt(caption_model='aoa', input_encoding_size=1024, rnn_size=1024, att_hid_size=512, num_heads=8,
                           multi_head_scale=1, use_multi_head=2, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
                           mean_feats=1, ctx_drop=1, dropout_aoa=0.3, drop_prob_lm=0.5, train_sample_n=n,
                           structure_loss_type='new_self_critical', structure_loss_weight=1.0, label_smoothing=0.2)
torch.manual_seed(1234)
model = models.setup(opt).to(dev)
flat = model.flatten_parameters_()
lw = LossWrapper(model, opt)
fc, att = synthetic.batch(B, seed=3, device=dev)
corpus = synthetic.corpus(500, seed=7)
df, ref_len = synthetic.document_frequency(corpus)
rewards.reset_scorer()
rewards.init_scorer((df, ref_len), device=dev)
gts = synthetic.corpus(B, seed=100)
labels, masks = synthetic.xe_labels(B, n=5, L=20)
labels, masks = labels.to(dev), masks.to(dev)


def step():
    out = lw(fc, att, labels, masks, None, gts, torch.arange(B), False, True, False)
    loss = out['loss'].mean()
    flat.zero_grad()
    loss.backward()
    flat.collect_grads()
    flat.adam_step(2e-5, clip_value=0.1)
    return out


for _ in range(4):
    out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 8
print('AoA new-self-critical bs10 x n5: %.2f ms/step = %.0f captions/s (struc_loss %.4f, reward %.4f)'
      % (dt * 1e3, B * n / dt, float(out['struc_loss'].mean().detach()), float(out['reward'].mean())), flush=True)
model.eval()
with torch.no_grad():
    for _ in range(2):
        seq, _ = model(fc, att, None, opt={'sample_method': 'beam_search', 'beam_size': 5, 'sample_n': 1}, mode='sample')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seq, _ = model(fc, att, None, opt={'sample_method': 'beam_search', 'beam_size': 5, 'sample_n': 1}, mode='sample')
    torch.cuda.synchronize()
    db = time.perf_counter() - t0
print('AoA beam-5 decode of %d images: %.1f ms (%.0f images/s)' % (B, db * 1e3, B / db), flush=True)
