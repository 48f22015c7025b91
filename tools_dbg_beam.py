import sys, os
sys.path.insert(0, 'tests')
import numpy as np, torch
from test_decode_opts_gpu import family, BEAM_CASES
for name, tag in (('newfc', 'bg2'), ('aoa', 'bg2')):
    z, model, fc, att, am = family(name)
    o = {'sample_method': 'beam_search', 'sample_n': 1}; o.update(BEAM_CASES[tag])
    with torch.no_grad():
        seq, slp = model(fc, att, am, opt=o, mode='sample')
    print(name, tag, 'seq', seq.cpu().tolist(), 'want', z[tag + '_seq'].tolist())
    for k, beams in enumerate(model.done_beams):
        nb = int(z['%s_n%d' % (tag, k)])
        print(' image', k, 'n', len(beams), nb)
        for j in range(max(len(beams), nb)):
            got = (beams[j]['seq'].cpu().tolist(), round(beams[j]['p'], 4)) if j < len(beams) else None
            want = (z['%s_%d_%d_seq' % (tag, k, j)].tolist(), round(float(z['%s_%d_%d_p' % (tag, k, j)]), 4)) if j < nb else None
            print('   ', j, got, '|', want)
