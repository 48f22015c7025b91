"""tools/eval.py on REAL, ragged-region data (VERDICT r4 missing #2).

The reference's eval loop moves ``data['att_masks']`` to the device and hands it to the loss forward, the decode and
``eval_split_n`` (eval_utils.py:157-171,198,236-274); its entry point evaluates on the real DataLoader (tools/eval.py:97,116).
Round 4's mirror passed ``None`` everywhere and only knew the synthetic fixed-36-region loader, so with adaptive 10-100 region
features every zero-padded row would have been attended.  These tests fail on that code:

* the committed ragged dataset ``tests/golden/loader_ds`` (3-7 regions per image) goes through ``eval.main`` /
  ``eval_split`` built from ``--input_json`` (FeatureLoader, with and without the HBM-resident store);
* predictions, loss, perplexity and entropy == the model called directly with the loader's masks, greedy and beam 5;
* and != the result without masks (the bug), so the test has teeth.
"""
import os
import sys

import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PKG = os.path.join(ROOT, 'imagecaptioning', 'pytorch_amd')
DS = os.path.join(GOLDEN, 'loader_ds')

FAMILY_ARGS = {
    'updown': ['--caption_model', 'updown'],
    'transformer': ['--caption_model', 'transformer', '--N_enc', '2', '--N_dec', '2', '--d_model', '16', '--d_ff', '32', '--num_att_heads', '2'],
    'aoa': ['--caption_model', 'aoa', '--num_heads', '2', '--num_layers', '2'],
}


def _opts(family, extra):
    sys.path.insert(0, PKG)
    from captioning.utils import opts
    argv = FAMILY_ARGS[family] + [
        '--input_json', os.path.join(DS, 'dataset.json'), '--input_label_h5', os.path.join(DS, 'labels.npz'),
        '--input_fc_dir', os.path.join(DS, 'fc'), '--input_att_dir', os.path.join(DS, 'att'),
        '--rnn_size', '16', '--input_encoding_size', '16', '--att_hid_size', '8', '--fc_feat_size', '6', '--att_feat_size', '6',
        '--batch_size', '3', '--seq_per_img', '2', '--num_images', '-1', '--split', 'val'] + extra
    return opts.parse_opt(argv)


def _direct(model, crit, loader, opt, use_masks):
    """what eval_split must compute, written against the model API alone (eval_utils.py:157-174)"""
    from imagecaptioning.pytorch_amd.tools import eval as E
    from captioning.utils import misc
    model.eval()
    loader.reset_iterator('val')
    out, losses, n, n_max = [], [], 0, None
    while n_max is None or n < n_max:
        d = loader.get_batch('val')
        n_max = d['bounds']['it_max']
        fc, att, labels, masks = (d[k].to(DEV) for k in ('fc_feats', 'att_feats', 'labels', 'masks'))
        am = d['att_masks'].to(DEV) if (use_masks and d['att_masks'] is not None) else None
        kw = E.eval_kwargs_of(opt)
        kw['sample_n'] = 1
        with torch.no_grad():
            losses.append(crit(model(fc, att, labels[..., :-1], am), labels[..., 1:], masks[..., 1:]).item())
            seq, lp = model(fc, att, am, mode='sample', opt=kw)
        steps = (seq > 0).to(lp).sum(1) + 1
        ent = -(torch.softmax(lp, 2) * lp).nan_to_num(0.0).sum(2).sum(1) / steps
        ppl = -lp.gather(2, seq.unsqueeze(2)).squeeze(2).sum(1) / steps
        for k, s in enumerate(misc.decode_sequence(model.vocab, seq)):
            out.append((d['infos'][k]['id'], s, ppl[k].item(), ent[k].item()))
        n += len(d['infos'])
    return sum(losses) / len(losses), out


@pytest.mark.parametrize('family', ['updown', 'transformer', 'aoa'])
@pytest.mark.parametrize('beam', [1, 5])
@pytest.mark.parametrize('resident', [1, 0])
def test_eval_split_carries_att_masks_of_ragged_batches(family, beam, resident):
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import eval as E
    from captioning import models
    from captioning.modules import losses
    opt = _opts(family, ['--beam_size', str(beam), '--resident_features', str(resident)])
    dev = torch.device(DEV)
    loader, opt.vocab = E.build_loader(opt, dev)
    assert opt.vocab_size == 20 and opt.seq_length == 5 and opt.max_length == 5
    first = loader.get_batch('val')
    assert first['att_masks'] is not None and first['att_feats'].shape[1] == 7            # regions 5 / 7 / 3, padded to 7
    assert float(first['att_masks'].sum()) == 15.0
    torch.manual_seed(77)
    model = models.setup(opt).to(dev)
    crit = losses.LanguageModelCriterion()
    loss, preds = E.eval_split(model, crit, loader, opt)
    assert model.training                                                                  # eval_utils.py:224-225
    assert [p['image_id'] for p in preds] == [1007, 1042, 1063]
    want_loss, want = _direct(model, crit, loader, opt, use_masks=True)
    assert abs(loss - want_loss) <= 1e-6
    for p, (iid, sent, ppl, ent) in zip(preds, want):
        assert p['image_id'] == iid and p['caption'] == sent
        assert abs(p['perplexity'] - ppl) <= 1e-5 and abs(p['entropy'] - ent) <= 1e-5
    # teeth: attending the zero-padded rows changes the numbers -- this is what round 4's eval_split computed
    bad_loss, bad = _direct(model, crit, loader, opt, use_masks=False)
    assert abs(bad_loss - loss) > 1e-4
    assert max(abs(p['perplexity'] - b[2]) for p, b in zip(preds, bad)) > 1e-4


@pytest.mark.parametrize('method', ['bs', 'sample', 'dbs', 'dgreedy'])
def test_eval_split_n_receives_the_masks(method):
    """eval_utils.py:198: eval_split_n gets [fc_feats, att_feats, att_masks, data]; spy on the model: every sampler call of the
    sample_n pass carries the batch's att_masks"""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import eval as E
    from captioning import models
    from captioning.modules import losses
    opt = _opts('updown', ['--sample_n', '2', '--sample_n_method', method, '--beam_size', '2'])
    dev = torch.device(DEV)
    loader, opt.vocab = E.build_loader(opt, dev)
    torch.manual_seed(78)
    model = models.setup(opt).to(dev)
    seen = []
    orig = model._sample

    def spy(fc, att, att_masks=None, opt={}):
        seen.append(None if att_masks is None else float(att_masks.sum()))
        return orig(fc, att, att_masks, opt)
    model._sample = spy
    E.eval_split(model, losses.LanguageModelCriterion(), loader, opt)
    assert len(seen) >= 2 and all(s == 15.0 for s in seen), seen
    assert len(model.n_predictions) == 3 * 2


def test_eval_main_on_the_real_loader_end_to_end(capsys):
    """tools/eval.py main(): --input_json picks the real-file loader (tools/eval.py:97-104); vocabulary, vocab_size and seq_length
    come from the dataset; a second ragged batch size (2 + 1 images: the single-image batch has no mask)"""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import eval as E
    opt = _opts('updown', ['--batch_size', '2', '--beam_size', '3'])
    loss, preds = E.main(opt)
    assert loss == loss and [p['image_id'] for p in preds] == [1007, 1042, 1063]
    words = set(opt.vocab.values())
    assert all(w in words for p in preds for w in p['caption'].split())
    assert 'loss:' in capsys.readouterr().out
