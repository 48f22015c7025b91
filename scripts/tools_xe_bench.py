#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json at their stated sizes: XE training step (teacher forcing, label smoothing off)
for UpDown bs64 (configs[1]), Transformer bs64 (configs[3]) and AoA bs10 new-self-critical-free XE (configs[4] model),
and greedy decoding throughput.  Not the headline metric (bench.py); reported in DESIGN.md section 6."""
import sys
import time

import torch

sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import synthetic                                   # noqa: E402
from imagecaptioning.pytorch_amd.captioning import models                          # noqa: E402
from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion   # noqa: E402

dev = torch.device('cuda:0')


def run(name, opt, B, n=5, L=20, steps=10, warm=4):
    torch.manual_seed(1234)
    model = models.setup(opt).to(dev)
    flat = model.flatten_parameters_()
    crit = LanguageModelCriterion()
    fc, att = synthetic.batch(B, seed=1234, device=dev)
    labels, masks = synthetic.xe_labels(B, n=n, L=L)
    labels, masks = labels.to(dev), masks.to(dev)
    model.train()

    def step():
        logp = model(fc, att, labels[..., :-1], None)
        loss = crit(logp, labels[..., 1:], masks[..., 1:])
        flat.zero_grad()
        loss.backward()
        flat.collect_grads()
        flat.adam_step(5e-4, clip_value=0.1)
        return loss

    for _ in range(warm):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            seq, _ = model(fc, att, None, opt={'sample_method': 'greedy', 'beam_size': 1}, mode='sample')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq, _ = model(fc, att, None, opt={'sample_method': 'greedy', 'beam_size': 1}, mode='sample')
        torch.cuda.synchronize()
        dg = time.perf_counter() - t0
    print('%-28s XE step %8.2f ms (%7.0f captions/s, loss %.3f)   greedy decode %7.2f ms (%6.0f images/s)'
          % (name, dt * 1e3, B * n / dt, float(loss.detach()), dg * 1e3, B / dg), flush=True)


which = sys.argv[1:] or ['updown', 'transformer', 'aoa']
if 'updown' in which:
    run('UpDown bs64 (configs[1])', synthetic.updown_opt(), 64)
if 'transformer' in which:
    o = synthetic.updown_opt(caption_model='transformer', input_encoding_size=512, rnn_size=2048, d_model=512, d_ff=2048, N_enc=6,
                             N_dec=6, num_att_heads=8, dropout=0.1, drop_prob_lm=0.5)
    run('Transformer bs64 (configs[3])', o, 64)
if 'aoa' in which:
    o = synthetic.updown_opt(caption_model='aoa', input_encoding_size=1024, rnn_size=1024, att_hid_size=512, num_heads=8,
                             multi_head_scale=1, use_multi_head=2, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
                             mean_feats=1, ctx_drop=1, dropout_aoa=0.3, drop_prob_lm=0.5)
    run('AoA bs10 (configs[4] model)', o, 10)
