import sys, torch
sys.path.insert(0, 'imagecaptioning/pytorch_amd')
from captioning.utils import opts, rewards
from imagecaptioning.pytorch_amd.tools import train as T
small = ['--caption_model', 'updown', '--rnn_size', '128', '--input_encoding_size', '128', '--att_hid_size', '64',
         '--fc_feat_size', '64', '--att_feat_size', '64', '--vocab_size', '100', '--synthetic_regions', '8', '--seq_length', '10',
         '--max_length', '10', '--batch_size', '16', '--seq_per_img', '5', '--synthetic_images', '64', '--losses_log_every', '50',
         '--checkpoint_path', '/tmp/capmi_learn', '--learning_rate', '0.002', '--drop_prob_lm', '0.1']
T.train(opts.parse_opt(small + ['--max_iters', '400', '--save_checkpoint_every', '400']))
rewards.reset_scorer()
T.train(opts.parse_opt(small + ['--max_iters', '300', '--self_critical_after', '0', '--train_sample_n', '5', '--start_from', '/tmp/capmi_learn',
                                 '--learning_rate', '0.0005']))
