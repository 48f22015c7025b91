"""tools/train.py and tools/eval.py mirrors run end to end on the HIP backend (synthetic data)."""
import os
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
PKG = os.path.join(ROOT, 'imagecaptioning', 'pytorch_amd')


def _opts(argv):
    sys.path.insert(0, PKG)
    from captioning.utils import opts
    return opts.parse_opt(argv)


def test_train_xe_then_scst_then_eval_beam(tmp_path):
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T, eval as E
    from captioning.utils import rewards
    small = ['--caption_model', 'updown', '--rnn_size', '64', '--input_encoding_size', '64', '--att_hid_size', '32',
             '--fc_feat_size', '48', '--att_feat_size', '48', '--vocab_size', '60', '--synthetic_regions', '7', '--seq_length', '8',
             '--max_length', '8', '--batch_size', '4', '--seq_per_img', '3', '--synthetic_images', '16', '--losses_log_every', '2',
             '--checkpoint_path', str(tmp_path)]
    l0 = T.train(_opts(small + ['--max_iters', '1']))
    l1 = T.train(_opts(small + ['--max_iters', '25', '--save_checkpoint_every', '25', '--learning_rate', '0.01',
                                '--learning_rate_decay_start', '0', '--learning_rate_decay_every', '2', '--val_every', '10',
                                '--val_images', '8', '--reduce_on_plateau', '0']))
    assert l1 < l0, 'XE loss should fall on a 16-image synthetic set (%.3f -> %.3f)' % (l0, l1)
    rewards.reset_scorer()
    # resumes at iteration 25 of the checkpoint (tools/train.py:50-66): three more, self-critical
    T.train(_opts(small + ['--max_iters', '28', '--self_critical_after', '0', '--train_sample_n', '3', '--start_from', str(tmp_path)]))
    rewards.reset_scorer()
    loss, preds = E.main(_opts(small + ['--beam_size', '3', '--sample_method', 'beam_search', '--num_images', '8',
                                         '--start_from', str(tmp_path)]))
    assert len(preds) == 8 and all(isinstance(p['caption'], str) for p in preds)
    assert loss == loss


def test_resume_restores_optimizer_schedule_counters_and_loader(tmp_path):
    """tools/train.py:50-119 + misc.save_checkpoint: model.pth, optimizer.pth (Adam moments, step count, schedule state) and
    infos (iter, epoch, loader position) -- a run stopped at iteration 4 and resumed to 8 ends with the SAME parameters as an
    uninterrupted run of 8 (warm-up schedule on, so a restart at iteration 0 would show)."""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    small = ['--caption_model', 'updown', '--rnn_size', '32', '--input_encoding_size', '32', '--att_hid_size', '16', '--fc_feat_size', '24',
             '--att_feat_size', '24', '--vocab_size', '40', '--synthetic_regions', '5', '--seq_length', '6', '--max_length', '6',
             '--batch_size', '4', '--seq_per_img', '2', '--synthetic_images', '12', '--use_warmup', '1', '--noamopt_warmup', '6',
             '--learning_rate', '0.01', '--drop_prob_lm', '0.5']
    a, b = tmp_path / 'a', tmp_path / 'b'
    T.train(_opts(small + ['--max_iters', '8', '--save_checkpoint_every', '8', '--checkpoint_path', str(a)]))
    T.train(_opts(small + ['--max_iters', '4', '--save_checkpoint_every', '4', '--checkpoint_path', str(b)]))
    import pickle
    infos = pickle.load(open(b / 'infos_capmi.pkl', 'rb'))
    assert infos['iter'] == 4 and infos['epoch'] == 1 and infos['loader_pos']['train'] == 4       # 12 images / 4 per batch
    osd = torch.load(b / 'optimizer.pth', weights_only=False)
    assert osd['flat']['step_count'] == 4 and float(osd['flat']['exp_avg_sq'].abs().max()) > 0
    T.train(_opts(small + ['--max_iters', '8', '--save_checkpoint_every', '8', '--checkpoint_path', str(b), '--start_from', str(b)]))
    pa, pb = torch.load(a / 'model.pth'), torch.load(b / 'model.pth')
    # (not bitwise: the embedding gradient is scattered with float atomics; a restart of the warm-up / of Adam's moments
    # would move the parameters by ~1e-2 with this learning rate)
    for k in pa:
        if k == 'core.attention.alpha_net.bias':
            continue      # its gradient is mathematically zero (softmax shift invariance): Adam normalises rounding noise to +-lr
        assert float((pa[k] - pb[k]).abs().max()) < 2e-5, (k, float((pa[k] - pb[k]).abs().max()))
    assert torch.load(b / 'optimizer.pth', weights_only=False)['flat']['step_count'] == 8


def test_unsupported_training_options_fail_loudly(tmp_path):
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    small = ['--rnn_size', '32', '--input_encoding_size', '32', '--att_hid_size', '16', '--fc_feat_size', '24', '--att_feat_size', '24',
             '--vocab_size', '40', '--synthetic_regions', '5', '--seq_length', '6', '--max_length', '6', '--batch_size', '4',
             '--seq_per_img', '2', '--synthetic_images', '8', '--max_iters', '1', '--checkpoint_path', str(tmp_path)]
    with pytest.raises(NotImplementedError):
        T.train(_opts(small + ['--grad_clip_mode', 'norm']))
    with pytest.raises(NotImplementedError):
        T.train(_opts(small + ['--optim', 'sgd']))
    # max_epochs of the reference configs is honoured (tools/train.py:279-280): 8 images / 4 per batch = 2 iterations per epoch
    opt = _opts(small[:-4] + ['--max_iters', '100', '--max_epochs', '2', '--checkpoint_path', str(tmp_path), '--save_checkpoint_every', '1000'])
    T.train(opt)
    import pickle
    assert pickle.load(open(tmp_path / 'infos_capmi.pkl', 'rb'))['iter'] == 4


def test_transformer_noam_schedule_runs(tmp_path):
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    small = ['--caption_model', 'transformer', '--d_model', '32', '--d_ff', '64', '--N_enc', '1', '--N_dec', '1',
             '--num_att_heads', '4', '--input_encoding_size', '32', '--rnn_size', '32', '--fc_feat_size', '24', '--att_feat_size', '24',
             '--vocab_size', '40', '--synthetic_regions', '5', '--seq_length', '6', '--max_length', '6', '--batch_size', '4',
             '--seq_per_img', '2', '--synthetic_images', '8', '--noamopt', '1', '--noamopt_warmup', '5', '--noamopt_factor', '1.0',
             '--checkpoint_path', str(tmp_path)]
    opt = _opts(small + ['--max_iters', '8'])
    loss = T.train(opt)
    assert loss == loss
    # NoamOpt.rate at the last step (8): past the 5-step warm-up, so step^-0.5 applies
    assert opt.current_lr == pytest.approx(32 ** -0.5 * 8 ** -0.5, rel=1e-9)


@pytest.mark.parametrize('family', ['transformer', 'aoa'])
def test_trainer_on_the_captured_step_equals_the_stepped_trainer(tmp_path, monkeypatch, family):
    """tools/train.py steps through graph_step.TrainStep (r6): for the Transformer and AoA families the iteration is captured into a
    hipGraph per input shape and replayed.  The SAME training run with CAPMI_GRAPH_STEP=0 (launch by launch) must end with the same
    loss to the last bit -- moving learning rate (Noam warm-up: capmi_step_set_lr outside the graph), dropout streams, Adam's bias
    corrections and all."""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    common = ['--fc_feat_size', '24', '--att_feat_size', '24', '--vocab_size', '40', '--synthetic_regions', '5', '--seq_length', '6',
              '--max_length', '6', '--batch_size', '4', '--seq_per_img', '2', '--synthetic_images', '8', '--max_iters', '10',
              '--losses_log_every', '1']
    if family == 'transformer':
        small = ['--caption_model', 'transformer', '--d_model', '32', '--d_ff', '64', '--N_enc', '1', '--N_dec', '1', '--num_att_heads', '4',
                 '--input_encoding_size', '32', '--rnn_size', '32', '--noamopt', '1', '--noamopt_warmup', '5', '--noamopt_factor', '1.0']
    else:
        small = ['--caption_model', 'aoa', '--rnn_size', '32', '--input_encoding_size', '32', '--att_hid_size', '16', '--num_heads', '4',
                 '--learning_rate', '1e-3']
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('CAPMI_GRAPH_STEP', mode)
        torch.manual_seed(0)
        opt = _opts(small + common + ['--checkpoint_path', str(tmp_path / ('g' + mode))])
        os.makedirs(opt.checkpoint_path, exist_ok=True)
        out[mode] = T.train(opt)
    assert out['1'] == out['1'] and out['1'] == out['0'], out


def test_device_prefetcher_delivers_the_same_batches_on_device():
    sys.path.insert(0, PKG)
    from captioning.data.synthetic_loader import SyntheticLoader
    from captioning.data.prefetch import DevicePrefetcher
    argv = ['--fc_feat_size', '16', '--att_feat_size', '16', '--vocab_size', '30', '--synthetic_regions', '4', '--seq_length', '6',
            '--max_length', '6', '--batch_size', '3', '--seq_per_img', '2', '--synthetic_images', '7']
    plain = SyntheticLoader(_opts(argv))
    pre = DevicePrefetcher(SyntheticLoader(_opts(argv)), 'cuda:0', depth=2)
    for _ in range(6):                                   # crosses the epoch boundary (7 images, batches of 3)
        a, b = plain.get_batch('train'), pre.get_batch('train')
        for k in ('fc_feats', 'att_feats', 'labels', 'masks'):
            assert b[k].is_cuda
            assert torch.equal(a[k], b[k].cpu()), k
        assert b['att_masks'] is None and a['bounds'] == b['bounds']
        assert all((x == y).all() for x, y in zip(a['gts'], b['gts']))
    assert pre.get_vocab() == plain.get_vocab()


def test_yaml_base_inheritance(tmp_path):
    sys.path.insert(0, PKG)
    from captioning.utils import config
    (tmp_path / 'base.yml').write_text('caption_model: updown\nrnn_size: 1000\nbatch_size: 10\n')
    (tmp_path / 'sc.yml').write_text('_BASE_: base.yml\nself_critical_after: 0\nbatch_size: 5\n')
    cfg = config.load(str(tmp_path / 'sc.yml'))
    assert cfg == {'caption_model': 'updown', 'rnn_size': 1000, 'batch_size': 5, 'self_critical_after': 0}


@pytest.mark.parametrize('model_args', [
    ['--caption_model', 'updown'],
    ['--caption_model', 'aoa', '--num_heads', '4', '--num_layers', '2'],
])
def test_eval_sample_n_methods_and_decode_flags(model_args):
    """eval_utils.eval_split_n (eval_utils.py:228-290): bs / sample / top-k / dbs / diverse sampling, plus the decode flags of
    the command line (decoding_constraint, block_trigrams, remove_bad_endings) reaching the sampler."""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import eval as E
    small = model_args + ['--rnn_size', '64', '--input_encoding_size', '64', '--att_hid_size', '32', '--fc_feat_size', '48',
                          '--att_feat_size', '48', '--vocab_size', '60', '--synthetic_regions', '7', '--seq_length', '8', '--max_length', '8',
                          '--batch_size', '4', '--seq_per_img', '3', '--synthetic_images', '16', '--num_images', '4']
    for method, extra in (('bs', []), ('sample', ['--temperature', '1.2']), ('top3', []), ('dbs', ['--beam_size', '2']),
                          ('dgreedy', ['--diversity_lambda', '2.0'])):
        import imagecaptioning.pytorch_amd.captioning.models  # noqa: F401
        opt = _opts(small + ['--sample_n', '3', '--sample_n_method', method, '--decoding_constraint', '1', '--block_trigrams', '1',
                             '--remove_bad_endings', '1'] + extra)
        captured = {}
        orig = E.eval_split

        def spy(model, crit, loader, o):
            r = orig(model, crit, loader, o)
            captured['n'] = model.n_predictions
            return r
        E.eval_split = spy
        try:
            loss, preds = E.main(opt)
        finally:
            E.eval_split = orig
        assert loss == loss and len(preds) >= 4
        n_preds = captured['n']
        assert len(n_preds) == 4 * 3, (method, len(n_preds))
        for p in preds + n_preds:
            words = p['caption'].split()
            assert all(a != b for a, b in zip(words, words[1:])), (method, p['caption'])      # decoding_constraint held


def test_train_with_scheduled_sampling(tmp_path):
    """tools/train.py:142-146: ss_prob rises with the epoch and reaches the model; the XE step runs with sampled inputs."""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    small = ['--caption_model', 'updown', '--rnn_size', '32', '--input_encoding_size', '32', '--att_hid_size', '16', '--fc_feat_size', '24',
             '--att_feat_size', '24', '--vocab_size', '40', '--synthetic_regions', '5', '--seq_length', '6', '--max_length', '6',
             '--batch_size', '4', '--seq_per_img', '2', '--synthetic_images', '8', '--checkpoint_path', str(tmp_path),
             '--scheduled_sampling_start', '0', '--scheduled_sampling_increase_every', '1', '--scheduled_sampling_increase_prob', '0.2',
             '--scheduled_sampling_max_prob', '0.5']
    opt = _opts(small + ['--max_iters', '10'])          # 2 iterations per epoch -> epochs 0..4
    loss = T.train(opt)
    assert loss == loss
    assert opt.ss_prob == pytest.approx(0.5)             # min(0.2 * 4, 0.5) at epoch 4


def test_train_drop_worst_xe_and_scst(tmp_path):
    """tools/train.py:160-165, 187-191: from drop_worst_after on, LossWrapper returns one loss per caption row (reduction 'none')
    and the rows with the highest loss are left out of the mean -- XE epochs first, then self-critical ones."""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    from captioning.modules import loss_wrapper as LW
    small = ['--caption_model', 'updown', '--rnn_size', '32', '--input_encoding_size', '32', '--att_hid_size', '16', '--fc_feat_size', '24',
             '--att_feat_size', '24', '--vocab_size', '40', '--synthetic_regions', '5', '--seq_length', '6', '--max_length', '6',
             '--batch_size', '4', '--seq_per_img', '2', '--synthetic_images', '8', '--checkpoint_path', str(tmp_path),
             '--drop_worst_after', '1', '--drop_worst_rate', '0.25', '--self_critical_after', '2', '--train_sample_n', '2']
    seen = []
    orig = LW.LossWrapper.forward

    def spy(self, *a):
        out = orig(self, *a)
        seen.append((a[-1], tuple(out['loss'].shape)))      # (drop_worst_flag, loss shape)
        return out
    LW.LossWrapper.forward = spy
    try:
        loss = T.train(_opts(small + ['--max_iters', '8']))          # 2 iterations per epoch -> epochs 0..3
    finally:
        LW.LossWrapper.forward = orig
    assert loss == loss
    assert [f for f, _ in seen] == [False, False] + [True] * 6
    assert all(sh == () for f, sh in seen if not f) and all(sh == (8,) for f, sh in seen if f)       # 4 images x 2 rows


def _file_dataset(tmp_path, n_img=40):
    import json
    import numpy as np
    rng = np.random.default_rng(0)
    (tmp_path / 'att').mkdir()
    labels, start, end, images = [], [], [], []
    for i in range(n_img):
        np.savez_compressed(tmp_path / 'att' / ('%d.npz' % i), feat=np.clip(rng.standard_normal((6, 24)), 0, None).astype(np.float32))
        start.append(len(labels) + 1)
        for _ in range(5):
            row = np.zeros(8, dtype=np.uint32)
            ln = int(rng.integers(3, 8))
            row[:ln] = rng.integers(1, 50, size=ln)
            labels.append(row)
        end.append(len(labels))
        images.append({'id': i, 'split': 'train'})
    (tmp_path / 'd.json').write_text(json.dumps({'images': images, 'ix_to_word': {str(i): 'w%d' % i for i in range(1, 50)}}))
    np.savez(tmp_path / 'l.npz', labels=np.stack(labels), label_start_ix=np.array(start, dtype=np.uint32),
             label_end_ix=np.array(end, dtype=np.uint32))
    return ['--caption_model', 'updown', '--rnn_size', '32', '--input_encoding_size', '32', '--att_hid_size', '16', '--att_feat_size', '24',
            '--fc_feat_size', '24', '--input_json', str(tmp_path / 'd.json'), '--input_label_h5', str(tmp_path / 'l.npz'),
            '--input_att_dir', str(tmp_path / 'att'), '--batch_size', '4', '--seq_per_img', '3', '--losses_log_every', '1000']


def test_unthrottled_host_with_the_resident_store_does_not_wedge(tmp_path, monkeypatch):
    """VERDICT r2 weak #9 / DESIGN r2 'known issue': with the HBM-resident feature store and a host that is never throttled
    (no per-iteration loss read-back, no early-exit waits) the process used to wedge after a few dozen iterations.  Root cause:
    the packed + cooked CIDEr-D references are allocated on the prefetcher's copy stream and read by the reward kernels on
    the main stream WITHOUT record_stream -- once the host runs several iterations ahead the caching allocator hands their
    blocks to the next side-stream pack while a queued reward kernel still walks them (prefetch.py now records the stream).
    500 un-throttled self-critical iterations must finish."""
    sys.path.insert(0, PKG)
    from imagecaptioning.pytorch_amd.tools import train as T
    from captioning.utils import rewards
    rewards.reset_scorer()
    monkeypatch.setenv('CAPMI_TRAIN_LAG', '-1')
    monkeypatch.setenv('CAPMI_EARLY_EXIT', '0')
    args = _file_dataset(tmp_path) + ['--self_critical_after', '0', '--train_sample_n', '3', '--max_iters', '500', '--max_epochs', '-1',
                                      '--resident_features', '1', '--checkpoint_path', str(tmp_path / 'ck')]
    loss = T.train(_opts(args))
    torch.cuda.synchronize()
    assert loss == loss
    rewards.reset_scorer()


def test_resume_inside_an_epoch_restores_order_and_rng_and_reads_a_reference_optimizer(tmp_path):
    """infos carry the shuffled order of the running epoch and both loader RNG streams (the reference saves
    its sampler's index_list / iter_counter, dataloader.py:376-405), so a run stopped INSIDE epoch 1 and resumed sees exactly the
    batches of the uninterrupted run; and an optimizer.pth written by the REFERENCE trainer (torch.optim.Adam.state_dict()) is
    converted into the flat moment buffers instead of raising KeyError."""
    sys.path.insert(0, PKG)
    import pickle
    from imagecaptioning.pytorch_amd.tools import train as T
    base = _file_dataset(tmp_path) + ['--learning_rate', '0.01', '--resident_features', '0']
    a, b = tmp_path / 'a', tmp_path / 'b'
    # 40 images / 4 per batch = 10 iterations per epoch: stop at 13 (inside epoch 1), resume to 17
    T.train(_opts(base + ['--max_iters', '17', '--save_checkpoint_every', '17', '--checkpoint_path', str(a)]))
    T.train(_opts(base + ['--max_iters', '13', '--save_checkpoint_every', '13', '--checkpoint_path', str(b)]))
    infos = pickle.load(open(b / 'infos_capmi.pkl', 'rb'))
    assert infos['epoch'] == 1 and len(infos['loader_order']['train']) == 40 and infos['loader_rng'] is not None
    assert set(infos['histories']) >= {'loss_history', 'lr_history', 'ss_prob_history'}
    T.train(_opts(base + ['--max_iters', '17', '--save_checkpoint_every', '17', '--checkpoint_path', str(b), '--start_from', str(b)]))
    pa, pb = torch.load(a / 'model.pth'), torch.load(b / 'model.pth')
    for k in pa:
        if k == 'core.attention.alpha_net.bias':
            continue
        assert float((pa[k] - pb[k]).abs().max()) < 5e-5, (k, float((pa[k] - pb[k]).abs().max()))
    # a reference-format optimizer.pth: torch Adam over the same parameters, two steps taken
    from captioning import models
    opt = _opts(base + ['--max_iters', '1'])
    opt.vocab_size, opt.seq_length, opt.max_length = 49, 8, 8
    opt.vocab = {str(i): 'w%d' % i for i in range(1, 50)}
    ref_model = models.setup(opt)
    ref_model.load_state_dict(torch.load(b / 'model.pth'))
    adam = torch.optim.Adam(ref_model.parameters(), lr=1e-3)
    for _ in range(2):
        for p in ref_model.parameters():
            p.grad = torch.full_like(p, 0.5)
        adam.step()
    c = tmp_path / 'c'
    c.mkdir()
    torch.save({k: v.detach().cpu() for k, v in ref_model.state_dict().items()}, c / 'model.pth')
    torch.save(adam.state_dict(), c / 'optimizer.pth')
    pickle.dump({'iter': 2, 'epoch': 0}, open(c / 'infos_capmi.pkl', 'wb'))
    T.train(_opts(base + ['--max_iters', '4', '--save_checkpoint_every', '4', '--checkpoint_path', str(c), '--start_from', str(c)]))
    osd = torch.load(c / 'optimizer.pth', weights_only=False)
    assert osd['flat']['step_count'] == 4                      # 2 converted + 2 new steps
