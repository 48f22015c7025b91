#!/usr/bin/env python3
"""Evaluation entrypoint on the MI355X backend -- the decode half of the reference's tools/eval.py:23-125 +
eval_utils.eval_split / eval_split_n (eval_utils.py:128-290): XE validation loss, decode with every sampler option of the
command line (beam / diverse beam search, sampling variants, decoding constraints), sample_n captions per image,
per-caption entropy and perplexity from seqLogprobs (:173-174), decoded strings.  (language_eval needs coco-caption + Java: out of scope.)

    python -m imagecaptioning.pytorch_amd.tools.eval --caption_model updown --beam_size 5 --num_images 20 [--start_from DIR]
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


SAMPLE_KEYS = ('sample_method', 'beam_size', 'temperature', 'suppress_UNK', 'length_penalty', 'group_size', 'diversity_lambda',
               'decoding_constraint', 'block_trigrams', 'remove_bad_endings', 'max_length')


def eval_kwargs_of(opt):
    """eval_utils.eval_split hands vars(opt) to the sampler (eval.py:103, eval_utils.py:169-171): every decode option of
    the command line reaches model._sample."""
    return {k: getattr(opt, k) for k in SAMPLE_KEYS if hasattr(opt, k)}


def eval_split_n(model, n_predictions, fc, att, att_masks, data, opt):
    """eval_utils.eval_split_n (eval_utils.py:228-290): sample_n captions per image by beam search ('bs'), sampling
    ('sample' / 'gumbel' / 'top<k|p>'), diverse beam search ('dbs') or diverse sampling ('d<method>')."""
    from captioning.utils import misc
    kw = eval_kwargs_of(opt)
    sample_n, method, beam_size = opt.sample_n, opt.sample_n_method, opt.beam_size
    with torch.no_grad():
        if method == 'bs':                                                  # :243-252
            kw.update(sample_n=1, beam_size=sample_n, group_size=1, sample_method='beam_search')
            model(fc, att, att_masks, opt=kw, mode='sample')
            for k in range(fc.shape[0]):
                for sent in misc.decode_sequence(model.vocab, _pad_stack([model.done_beams[k][i]['seq'] for i in range(sample_n)])):
                    n_predictions.append({'image_id': data['infos'][k]['id'], 'caption': sent})
        elif method in ('sample', 'gumbel') or method.startswith('top'):   # :254-264
            kw.update(sample_n=sample_n, sample_method=method, beam_size=1)
            seq, logp = model(fc, att, att_masks, opt=kw, mode='sample')
            ppl = -logp.gather(2, seq.unsqueeze(2)).squeeze(2).sum(1) / ((seq > 0).to(logp).sum(1) + 1)
            for k, sent in enumerate(misc.decode_sequence(model.vocab, seq)):
                n_predictions.append({'image_id': data['infos'][k // sample_n]['id'], 'caption': sent, 'perplexity': ppl[k].item()})
        elif method == 'dbs':                                               # :265-274
            kw.update(beam_size=sample_n * beam_size, group_size=sample_n, sample_method='beam_search', sample_n=1)
            model(fc, att, att_masks, opt=kw, mode='sample')
            for k in range(fc.shape[0]):
                picks = [model.done_beams[k][i]['seq'] for i in range(0, sample_n * beam_size, beam_size)]
                for sent in misc.decode_sequence(model.vocab, _pad_stack(picks)):
                    n_predictions.append({'image_id': data['infos'][k]['id'], 'caption': sent})
        else:                                                               # :275-283 diverse sampling, 'd' + method
            kw.update(sample_method=method[1:], group_size=sample_n, beam_size=1)
            seq, _ = model(fc, att, att_masks, opt=kw, mode='sample')
            for k, sent in enumerate(misc.decode_sequence(model.vocab, seq)):
                n_predictions.append({'image_id': data['infos'][k // sample_n]['id'], 'caption': sent})


def _pad_stack(seqs):
    """torch.stack of the beams' token rows; finished beams are shorter than seq_length here (the reference stacks rows that
    all ran to the same length only when none ended early), so pad with the end token."""
    ln = max(s.shape[0] for s in seqs)
    return torch.stack([torch.cat([s, s.new_zeros(ln - s.shape[0])]) for s in seqs])


def eval_split(model, crit, loader, opt):
    from captioning.utils import misc
    dev = next(model.parameters()).device
    model.eval()
    n, loss_sum, loss_n, preds, n_preds = 0, 0.0, 0, [], []
    split = getattr(opt, 'split', 'val')
    if hasattr(loader, 'reset_iterator'):
        loader.reset_iterator(split)                                                                           # eval_utils.py:145
    num_images = opt.num_images
    while num_images < 0 or n < num_images:
        data = loader.get_batch(split)
        # eval_utils.py:200-207: ix1 = min(it_max, num_images) -- never more than the split holds (a larger request would make the
        # non-wrapping loader start the split over and every image would be predicted twice)
        num_images = data['bounds']['it_max'] if num_images < 0 else min(num_images, data['bounds']['it_max'])
        # eval_utils.py:157-159: fc_feats, att_feats, labels, masks AND att_masks go to the device -- with 10..100 adaptive
        # regions per image the padded rows of att_feats must stay out of the attention (None when the batch is not ragged)
        fc, att, labels, masks = (data[k].to(dev) for k in ('fc_feats', 'att_feats', 'labels', 'masks'))
        att_masks = None if data.get('att_masks') is None else data['att_masks'].to(dev)
        kw = eval_kwargs_of(opt)
        kw['sample_n'] = 1                                                                                     # :169-170
        with torch.no_grad():
            loss = crit(model(fc, att, labels[..., :-1], att_masks), labels[..., 1:], masks[..., 1:]).item()  # eval_utils.py:163
            seq, seq_logp = model(fc, att, att_masks, mode='sample', opt=kw)                                   # :171
        loss_sum += loss
        loss_n += 1
        if seq_logp.dim() == 3:
            # eval_utils.py:173-174: sums over ALL L steps (rows after the end are zero) over (#tokens + 1); the only deviation:
            # 0 * -inf of a constrained token counts as 0 instead of making the whole caption's entropy NaN.  r6: one pass of
            # capmi_caption_stats over the log-probs the decode returned -- the ATen formula built three dense [N, L, V1] temporaries
            from imagecaptioning.pytorch_amd import ops
            entropy, perplexity = ops.caption_stats(seq_logp.contiguous(), seq.contiguous())
        else:                                  # _diverse_sample returns the chosen tokens' log-probs only (AttModel.py:449)
            entropy = perplexity = torch.full((seq.shape[0],), float('nan'))
        if opt.beam_size > 1 and getattr(opt, 'verbose_beam', 0):                                              # :177-181
            for i in range(fc.shape[0]):
                print('\n'.join(misc.decode_sequence(model.vocab, b['seq'].unsqueeze(0))[0] for b in model.done_beams[i]))
                print('--' * 10)
        sents = misc.decode_sequence(model.vocab, seq)
        rows_per_image = max(1, len(sents) // len(data['infos']))
        for k, s in enumerate(sents):
            preds.append({'image_id': data['infos'][k // rows_per_image]['id'], 'caption': s, 'perplexity': perplexity[k].item(),
                          'entropy': entropy[k].item()})
        if opt.sample_n > 1:                                                                                   # :199-200
            eval_split_n(model, n_preds, fc, att, att_masks, data, opt)                                       # :198
        n += len(data['infos'])
    if n_preds and 'perplexity' in n_preds[0]:
        n_preds = sorted(n_preds, key=lambda x: x['perplexity'])                                               # :217-218
    model.n_predictions = n_preds
    model.train()                                                                                              # :224-225
    return loss_sum / max(loss_n, 1), preds[:num_images * max(1, len(preds) // max(n, 1))]


def build_loader(opt, dev):
    """tools/eval.py:97-104: the reference evaluates on its DataLoader (precomputed bottom-up features, 10-100 regions per image
    => ragged batches with att_masks).  --input_json selects the same real-file loader tools/train.py uses (FeatureLoader, kept
    resident in HBM unless --resident_features 0); without it the synthetic fixed-36-region loader stands in."""
    if getattr(opt, 'input_json', ''):
        from captioning.data.feature_loader import FeatureLoader
        loader = FeatureLoader(opt)
        opt.vocab_size, opt.seq_length = loader.vocab_size, loader.seq_length
        if not getattr(opt, 'max_length', None) or opt.max_length > opt.seq_length:
            opt.max_length = opt.seq_length
        vocab = loader.get_vocab()
        if getattr(opt, 'resident_features', 1):
            from captioning.data.resident import ResidentFeatures
            budget = int(opt.resident_budget_gb * (1 << 30)) if getattr(opt, 'resident_budget_gb', 0) > 0 else None
            loader = ResidentFeatures(loader, dev, budget_bytes=budget)
        return loader, vocab
    from captioning.data.synthetic_loader import SyntheticLoader
    loader = SyntheticLoader(opt)
    return loader, loader.get_vocab()


def main(opt):
    from captioning import models
    from captioning.modules import losses
    dev = torch.device(opt.device if opt.device != 'cuda' else 'cuda:0')
    loader, opt.vocab = build_loader(opt, dev)
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
