"""Round-4 sanity run of tools/train.py on the paths this round touched: UpDown XE -> SCST (fused select + GEMM launch), AoA
new-self-critical and Transformer XE with FLATTENED parameters (fused q|k|v projections, deferred gradient reductions), UpDown with the
max_margin structure loss (raw-logit rollouts).  Prints the first / last smoothed losses; they must fall (XE) or the reward rise."""
import sys, torch
sys.path.insert(0, 'imagecaptioning/pytorch_amd')
from captioning.utils import opts, rewards
from imagecaptioning.pytorch_amd.tools import train as T
base = ['--rnn_size', '128', '--input_encoding_size', '128', '--att_hid_size', '64', '--fc_feat_size', '64', '--att_feat_size', '64',
        '--vocab_size', '100', '--synthetic_regions', '8', '--seq_length', '10', '--max_length', '10', '--batch_size', '16',
        '--seq_per_img', '5', '--synthetic_images', '64', '--losses_log_every', '100', '--drop_prob_lm', '0.1']
runs = [
    ('updown XE', ['--caption_model', 'updown', '--max_iters', '300', '--learning_rate', '0.002', '--checkpoint_path', '/tmp/r4_u', '--save_checkpoint_every', '300']),
    ('updown SCST', ['--caption_model', 'updown', '--max_iters', '500', '--self_critical_after', '0', '--train_sample_n', '5', '--start_from', '/tmp/r4_u', '--learning_rate', '0.0005', '--checkpoint_path', '/tmp/r4_u2']),
    ('updown max_margin', ['--caption_model', 'updown', '--max_iters', '400', '--structure_after', '0', '--structure_loss_type', 'max_margin', '--train_sample_n', '5', '--start_from', '/tmp/r4_u', '--learning_rate', '0.0005', '--checkpoint_path', '/tmp/r4_u3']),
    ('transformer XE', ['--caption_model', 'transformer', '--d_model', '64', '--d_ff', '128', '--N_enc', '2', '--N_dec', '2', '--num_att_heads', '4', '--max_iters', '300', '--learning_rate', '0.002', '--checkpoint_path', '/tmp/r4_t']),
    ('aoa XE', ['--caption_model', 'aoa', '--num_heads', '4', '--max_iters', '300', '--learning_rate', '0.002', '--checkpoint_path', '/tmp/r4_a', '--save_checkpoint_every', '300']),
    ('aoa new_self_critical', ['--caption_model', 'aoa', '--num_heads', '4', '--max_iters', '450', '--structure_after', '0', '--structure_loss_type', 'new_self_critical', '--train_sample_n', '5', '--start_from', '/tmp/r4_a', '--learning_rate', '0.0005', '--checkpoint_path', '/tmp/r4_a2']),
]
for name, extra in runs:
    print('=====', name, flush=True)
    rewards.reset_scorer()
    T.train(opts.parse_opt(base + extra))
