"""Synthetic workload of BASELINE.json / SURVEY.md 8d: 36x2048 bottom-up features, UpDown sizes from
configs/updown/updown.yml:13-16, vocab 9487, 5 references per image, Zipfian document frequencies."""
import argparse

import numpy as np
import torch

VOCAB = 9487


def updown_opt(**over):
    o = argparse.Namespace(
        caption_model='updown', vocab_size=VOCAB, input_encoding_size=1000, rnn_size=1000, num_layers=1,
        drop_prob_lm=0.5, seq_length=20, max_length=20, fc_feat_size=2048, att_feat_size=2048, att_hid_size=512,
        use_bn=0, logit_layers=1, rnn_type='lstm',
        vocab={str(i): ('w%d' % i) for i in range(1, VOCAB)} | {str(VOCAB): 'UNK'},
        # training knobs (opts.py defaults / configs/updown/updown_sc.yml)
        label_smoothing=0, structure_loss_type=None, structure_loss_weight=0, use_ppo=0,
        sc_sample_method='greedy', sc_beam_size=1, train_sample_method='sample', train_beam_size=1, train_sample_n=5,
        cider_reward_weight=1, bleu_reward_weight=0, learning_rate=5e-4, optim_alpha=0.9, optim_beta=0.999,
        optim_epsilon=1e-8, weight_decay=0, grad_clip_mode='value', grad_clip_value=0.1)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def zipf_rows(rng, rows, width, vocab=VOCAB, a=1.2, min_len=8):
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    p = ranks ** (-a)
    p /= p.sum()
    arr = np.zeros((rows, width), dtype=np.uint32)
    for j in range(rows):
        ln = int(rng.integers(min(min_len, width), width + 1))
        arr[j, :ln] = rng.choice(vocab, size=ln, p=p) + 1
    return arr


def corpus(num_images, refs_per_image=5, width=20, seed=1234):
    rng = np.random.default_rng(seed)
    return [zipf_rows(rng, refs_per_image, width) for _ in range(num_images)]


def document_frequency(ref_sets):
    """{ngram tuple -> #images containing it}, ref_len (scripts/prepro_ngrams.py:17-22 semantics)."""
    df = {}
    for refs in ref_sets:
        seen = set()
        for row in refs:
            toks = []
            for v in row:
                toks.append(int(v))
                if v == 0:
                    break
            for k in range(1, 5):
                for i in range(len(toks) - k + 1):
                    seen.add(tuple(toks[i:i + k]))
        for g in seen:
            df[g] = df.get(g, 0.0) + 1.0
    return df, len(ref_sets)


def batch(B, K=36, F=2048, seed=1234, device='cpu'):
    g = torch.Generator().manual_seed(seed)
    fc = (torch.randn(B, F, generator=g) * 0.5).clamp_min(0)
    att = (torch.randn(B, K, F, generator=g) * 0.5).clamp_min(0)
    return fc.to(device), att.to(device)


def xe_labels(B, n=5, L=20, seed=1234):
    """labels [B,n,L+2] (col 0 and last = 0), masks per dataloader.py:245-249."""
    rng = np.random.default_rng(seed)
    labels = np.zeros((B, n, L + 2), dtype=np.int64)
    masks = np.zeros((B, n, L + 2), dtype=np.float32)
    for b in range(B):
        rows = zipf_rows(rng, n, L)
        labels[b, :, 1:L + 1] = rows
        for j in range(n):
            masks[b, j, :int((rows[j] > 0).sum()) + 2] = 1
    return torch.from_numpy(labels), torch.from_numpy(masks)
