#!/usr/bin/env python3
"""Evaluation entrypoint on the MI355X backend -- the decode half of the reference's tools/eval.py:23-125 +
eval_utils.eval_split (eval_utils.py:128-226): XE validation loss, greedy / beam decode, per-caption entropy and
perplexity from seqLogprobs (:173-174), decoded strings.  (language_eval needs coco-caption + Java: out of scope.)

    python -m imagecaptioning.pytorch_amd.tools.eval --caption_model updown --beam_size 5 --num_images 20 [--start_from DIR]
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def eval_split(model, crit, loader, opt):
    from captioning.utils import misc
    dev = next(model.parameters()).device
    model.eval()
    n, loss_sum, loss_n, preds = 0, 0.0, 0, []
    while n < opt.num_images:
        data = loader.get_batch('val')
        fc, att, labels, masks = (data[k].to(dev) for k in ('fc_feats', 'att_feats', 'labels', 'masks'))
        with torch.no_grad():
            loss = crit(model(fc, att, labels[..., :-1], None), labels[..., 1:], masks[..., 1:]).item()       # eval_utils.py:163
            seq, seq_logp = model(fc, att, None, mode='sample',
                                  opt={'sample_method': opt.sample_method, 'beam_size': opt.beam_size, 'sample_n': 1,
                                       'temperature': opt.temperature, 'suppress_UNK': opt.suppress_UNK,
                                       'length_penalty': opt.length_penalty})                                 # :171
        loss_sum += loss
        loss_n += 1
        mask = (seq > 0).to(seq_logp)
        mask = torch.cat([mask.new_ones(mask.shape[0], 1), mask[:, :-1]], 1)
        entropy = -(torch.softmax(seq_logp, 2) * seq_logp).sum(2)                                              # :173
        entropy = (entropy * mask).sum(1) / mask.sum(1)
        perplexity = -(seq_logp.gather(2, seq.unsqueeze(2)).squeeze(2) * mask).sum(1) / mask.sum(1)           # :174
        sents = misc.decode_sequence(model.vocab, seq)
        for k, s in enumerate(sents):
            preds.append({'image_id': data['infos'][k]['id'], 'caption': s, 'perplexity': perplexity[k].item(),
                          'entropy': entropy[k].item()})
        n += len(sents)
    return loss_sum / max(loss_n, 1), preds[:opt.num_images]


def main(opt):
    from captioning import models
    from captioning.data.synthetic_loader import SyntheticLoader
    from captioning.modules import losses
    dev = torch.device(opt.device if opt.device != 'cuda' else 'cuda:0')
    loader = SyntheticLoader(opt)
    opt.vocab = loader.get_vocab()
    torch.manual_seed(1234)
    model = models.setup(opt).to(dev)
    if opt.start_from:
        model.load_state_dict(torch.load(os.path.join(opt.start_from, 'model.pth'), map_location=dev))
    crit = losses.LabelSmoothing(smoothing=opt.label_smoothing) if opt.label_smoothing > 0 else losses.LanguageModelCriterion()
    loss, preds = eval_split(model, crit, loader, opt)
    print('loss: ', loss)
    for p in preds[:5]:
        print('image %s: %s' % (p['image_id'], p['caption']))
    return loss, preds


if __name__ == '__main__':
    from captioning.utils import opts
    main(opts.parse_opt())
