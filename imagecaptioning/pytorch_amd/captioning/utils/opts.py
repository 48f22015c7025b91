"""Option parsing for the entrypoints: the subset of the reference's flags (captioning/utils/opts.py:18-277) that
the hot path reads, with the reference's defaults; ``--cfg`` yaml files (with ``_BASE_``) and CLI overrides are
applied in the reference's order (defaults < yaml < CLI)."""
import argparse

from . import config

DEFAULTS = dict(
    caption_model='updown', rnn_size=512, num_layers=1, input_encoding_size=512, att_hid_size=512, fc_feat_size=2048,
    att_feat_size=2048, logit_layers=1, use_bn=0, drop_prob_lm=0.5, seq_length=20, max_length=20, seq_per_img=5, batch_size=10,
    max_epochs=-1, max_iters=100, learning_rate=4e-4, optim='adam', optim_alpha=0.9, optim_beta=0.999, optim_epsilon=1e-8,
    weight_decay=0.0, grad_clip_mode='value', grad_clip_value=0.1, label_smoothing=0.0, self_critical_after=-1, structure_after=-1,
    structure_loss_weight=1.0, structure_loss_type='seqnll', train_sample_n=16, train_sample_method='sample', train_beam_size=1,
    sc_sample_method='greedy', sc_beam_size=1, cider_reward_weight=1.0, bleu_reward_weight=0.0, cached_tokens='coco-train-idxs',
    use_ppo=0, entropy_reward_weight=0.0, self_cider_reward_weight=0.0, struc_use_logsoftmax=0, drop_worst_after=-1, drop_worst_rate=0.0,
    learning_rate_decay_start=-1, learning_rate_decay_every=3, learning_rate_decay_rate=0.8, noamopt=0, noamopt_warmup=2000,
    noamopt_factor=1.0, use_warmup=0, reduce_on_plateau=0, reduce_on_plateau_factor=0.5, reduce_on_plateau_patience=3,
    scheduled_sampling_start=-1, scheduled_sampling_increase_every=5, scheduled_sampling_increase_prob=0.05,
    scheduled_sampling_max_prob=0.25, val_every=0, val_images=0,
    checkpoint_path='log_capmi', id='capmi', save_checkpoint_every=0, losses_log_every=10, start_from=None, seed=1234,
    # transformer / aoa
    N_enc=6, N_dec=6, d_model=512, d_ff=2048, num_att_heads=8, dropout=0.1, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
    use_multi_head=2, num_heads=8, multi_head_scale=1, mean_feats=1, ctx_drop=1, dropout_aoa=0.3,
    # eval
    beam_size=1, sample_method='greedy', temperature=1.0, suppress_UNK=1, length_penalty='', num_images=20, device='cuda',
    group_size=1, diversity_lambda=0.5, decoding_constraint=0, block_trigrams=0, remove_bad_endings=0, sample_n=1,
    sample_n_method='sample', verbose_beam=0,                                # opts.py:288-330 add_eval_sample_opts
    split='test',                                                            # opts.py:314 add_eval_options
    # data (synthetic only: the reference's h5/lmdb loaders are outside the hot path)
    input_synthetic=1, vocab_size=9487, synthetic_regions=36, synthetic_images=200,
    # real precomputed features (captioning/data/feature_loader.py; opts.py:23-37 of the reference)
    input_json='', input_label_h5='', input_fc_dir='', input_att_dir='', use_fc=1, norm_att_feat=0, train_only=0, resident_features=1, resident_budget_gb=0.0,
)


def parse_opt(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', type=str, default=None)
    for k, v in DEFAULTS.items():
        ap.add_argument('--' + k, type=(type(v) if v is not None else str), default=None)
    ns = ap.parse_args(argv)
    opt = dict(DEFAULTS)
    if ns.cfg:
        for k, v in config.load(ns.cfg).items():
            opt[k] = v
    for k in DEFAULTS:
        v = getattr(ns, k)
        if v is not None:
            opt[k] = v
    opt['cfg'] = ns.cfg
    o = argparse.Namespace(**opt)
    if o.max_length is None:
        o.max_length = o.seq_length
    return o
