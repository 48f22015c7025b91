#!/usr/bin/env python3
"""CPU baseline "reference": the IMPORTED reference modules (/root/reference on PYTHONPATH, build container only) running one
full UpDown SCST iteration of BASELINE configs[2] -- captioning.models.setup('updown') + the LossWrapper self-critical flow
(loss_wrapper.py:56-73: eval-mode greedy rollout, train-mode sampled rollout x5, get_self_critical_reward, RewardCriterion) +
backward + clip_grad_value_(0.1) + Adam -- on the same synthetic batch bench.py uses.  The external CIDEr-D package is absent
from the checkout, so rewards.CiderD_scorer is the stub of tests/golden/make_golden.py (upstream compute_score interface over
oracle/ciderd.py).  Writes profiles/r02_cpu_reference.json, which bench.py reports next to the on-box timing of the port.

    PYTHONDONTWRITEBYTECODE=1 python scripts/time_reference_cpu.py [threads]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get('CAPMI_REFERENCE', '/root/reference'))
sys.dont_write_bytecode = True


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    import captioning.models as models                     # the reference
    import captioning.utils.rewards as R
    from captioning.modules.loss_wrapper import LossWrapper
    from imagecaptioning.pytorch_amd import synthetic
    from oracle import ciderd as C

    opt = synthetic.updown_opt()
    opt.vocab = synthetic.updown_opt().vocab
    torch.manual_seed(1234)
    model = models.setup(opt)
    corpus = synthetic.corpus(10000, seed=7)
    df, ref_len = synthetic.document_frequency(corpus)
    oracle = C.CiderD(df, ref_len)

    class Stub:                                             # upstream CiderD.compute_score(gts, res) interface
        def compute_score(self, gts, res):
            hyps = [[int(t) for t in r['caption'][0].split()] for r in res]
            refs = [[[int(t) for t in s.split()] for s in gts[r['image_id']]] for r in res]
            return oracle.compute_score(hyps, refs)
    R.CiderD_scorer = Stub()
    lw = LossWrapper(model, opt)
    B = 10
    fc, att = synthetic.batch(B, seed=1234)
    gts = synthetic.corpus(B, seed=100)
    optim = torch.optim.Adam(model.parameters(), lr=opt.learning_rate, betas=(opt.optim_alpha, opt.optim_beta),
                             eps=opt.optim_epsilon)

    def step():
        model.train()
        with contextlib.redirect_stdout(io.StringIO()):    # the reference prints the mean CIDEr score
            out = lw(fc, att, None, None, None, gts, torch.arange(B), True, False, False)
        loss = out['loss'].mean()
        optim.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), opt.grad_clip_value)   # tools/train.py:194
        optim.step()
        return float(loss)

    for _ in range(2):
        step()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    sec = float(np.median(times))
    res = {'kind': 'reference', 'where': 'build container (no GPU box access to /root/reference)', 'cores': threads,
           'host_cpu_count': os.cpu_count(), 'sec_per_iteration': round(sec, 3),
           'value': round(B * opt.train_sample_n / sec, 2), 'unit': 'captions/s',
           'sample': '5 timed SCST iterations (bs10 x n5, L=20) of the imported reference LossWrapper flow after 2 warm-up, median; '
                     'torch %s fp32, %d threads; DF table from 10000 synthetic reference sets' % (torch.__version__, threads)}
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    with open(os.path.join(ROOT, 'profiles', 'r02_cpu_reference.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
