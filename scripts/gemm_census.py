#!/usr/bin/env python3
"""Census of the fat-GEMM shapes of one Transformer XE step: CAPMI_GEMM_LOG=1 python scripts/gemm_census.py 2> log; aggregates the log."""
import collections
import re
import sys

if len(sys.argv) > 1:
    c = collections.Counter()
    for line in open(sys.argv[1]):
        m = re.match(r'capmi_gemm (.*)', line)
        if m:
            c[m.group(1)] += 1
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print('%5d  %s' % (v, k))
    sys.exit(0)
import torch
sys.path.insert(0, '.')
from imagecaptioning.pytorch_amd import synthetic                                   # noqa: E402
from imagecaptioning.pytorch_amd.captioning import models                          # noqa: E402
from imagecaptioning.pytorch_amd.captioning.modules.losses import LanguageModelCriterion   # noqa: E402
dev = torch.device('cuda:0')
o = synthetic.updown_opt(caption_model='transformer', input_encoding_size=512, rnn_size=2048, d_model=512, d_ff=2048, N_enc=6,
                         N_dec=6, num_att_heads=8, dropout=0.1, drop_prob_lm=0.5)
torch.manual_seed(1234)
model = models.setup(o).to(dev)
flat = model.flatten_parameters_()
crit = LanguageModelCriterion()
fc, att = synthetic.batch(64, seed=1234, device=dev)
labels, masks = synthetic.xe_labels(64, n=5, L=20)
labels, masks = labels.to(dev), masks.to(dev)
model.train()
logp = model(fc, att, labels[..., :-1], None)
loss = crit(logp, labels[..., 1:], masks[..., 1:])
flat.zero_grad()
loss.backward()
torch.cuda.synchronize()
