#!/usr/bin/env python3
"""Training entrypoint on the MI355X backend -- the control flow of the reference's tools/train.py:32-292 for the
hot path: learning-rate decay / warm-up / Noam / reduce-on-plateau (:94-102, :132-141, :171-173, :253-256), XE /
self-critical / new-self-critical scheduling (:144-161), LossWrapper call (:185), backward, value clip + Adam
(:193-196), `time/batch` print (:198-208), periodic validation loss and checkpoint (:228-285), bucketed flat-gradient
RCCL all-reduce overlapped with the backward when launched with torch.distributed.run.

    python -m imagecaptioning.pytorch_amd.tools.train --caption_model updown --rnn_size 1000 --input_encoding_size 1000 \
        --self_critical_after 0 --train_sample_n 5 --max_iters 50
    python -m imagecaptioning.pytorch_amd.tools.train --cfg /path/to/reference/configs/updown/updown.yml --max_iters 20
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))           # so that `captioning` resolves to the mirror package


def validation_loss(lw_model, loader, opt, dev, world=1):
    """XE loss over `val_images` images of the val split, teacher forced, eval mode (eval_utils.py:150-160).  With `world`
    data-parallel ranks the val split is partitioned like the train split: each rank scores at most its share
    (val_images / world, never more than its partition holds) and returns (sum of per-image losses, images) so that the caller
    can all-reduce BOTH and every image counts once, whatever the partition sizes.  A rank whose partition is empty (val split
    smaller than world) contributes (0, 0)."""
    from captioning.data.feature_loader import EmptySplit
    model = lw_model.model
    model.eval()
    tot, n = 0.0, 0
    if hasattr(loader, 'reset_iterator'):
        loader.reset_iterator('val')                       # eval_utils.py:145: every evaluation starts at the top of the split
    want = max(opt.val_images, opt.batch_size)
    if world > 1:
        want = max(1, -(-want // world))
    with torch.no_grad():
        while n < want:
            try:
                data = loader.get_batch('val')
            except EmptySplit:                             # this rank's part of the split is empty; any other error propagates
                break
            it_max = data['bounds'].get('it_max')
            if it_max is not None:
                want = min(want, max(int(it_max), 1))      # eval_utils.py:200-207: never more than the split holds
            fc, att, labels, masks = (data[k].to(dev) for k in ('fc_feats', 'att_feats', 'labels', 'masks'))
            att_masks = None if data['att_masks'] is None else data['att_masks'].to(dev)
            logp = model(fc, att, labels[..., :-1], att_masks)
            tot += float(lw_model.crit(logp, labels[..., 1:], masks[..., 1:])) * fc.shape[0]
            n += fc.shape[0]
    model.train()
    return tot, n


def train(opt):
    import torch.distributed as dist
    from captioning import models
    from captioning.data.synthetic_loader import SyntheticLoader
    from captioning.data.prefetch import DevicePrefetcher
    from captioning.modules.loss_wrapper import LossWrapper
    from captioning.utils import rewards, misc
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if 'OMP_NUM_THREADS' not in os.environ:
        # the host only assembles batches and issues launches: torch's default of 128 OpenMP threads on this 256-core host
        # spin after every incidental CPU op and starve the decode threads (36 vs 8 ms per iteration, scripts/train_e2e.sh)
        torch.set_num_threads(4)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', rank=rank, world_size=world)
        dist.barrier(device_ids=[local])              # RCCL communicator created on the main thread, before any backward
    if opt.optim != 'adam':
        raise NotImplementedError("optim %r: the fused flat-buffer step implements Adam (misc.py:125-126, every BASELINE "
                                  "config); other optimizers of misc.build_optimizer are out of scope" % opt.optim)
    if opt.grad_clip_mode not in ('value',) and opt.grad_clip_value > 0:
        raise NotImplementedError("grad_clip_mode %r: only clip_grad_value_ (train.py:194-195, the default) is fused into the "
                                  "Adam kernel" % opt.grad_clip_mode)
    # Every rank draws its own images -- as a PARTITION of one shuffled pass: the loaders share `seed` and take
    # every world-th element of the permutation (tools/train_pl.py:60-73 + Lightning's DistributedSampler; per-GPU batch :459-460)
    if opt.input_json:                                # precomputed bottom-up features + labels (variable region counts)
        from captioning.data.feature_loader import FeatureLoader
        loader = FeatureLoader(opt, rank=rank, world=world)
        opt.vocab_size, opt.seq_length = loader.vocab_size, loader.seq_length
        if not getattr(opt, 'max_length', None) or opt.max_length > opt.seq_length:
            opt.max_length = opt.seq_length
    else:
        loader = SyntheticLoader(opt, rank=rank, world=world)
    opt.vocab = loader.get_vocab()
    if opt.input_json and getattr(opt, 'resident_features', 1):
        # the whole feature set lives in HBM after its first epoch (36 GB for COCO at 36 regions; budget: half the device memory)
        from captioning.data.resident import ResidentFeatures
        loader = ResidentFeatures(loader, dev, budget_bytes=int(opt.resident_budget_gb * (1 << 30)) if opt.resident_budget_gb > 0 else None)
    loader = DevicePrefetcher(loader, dev)             # batches arrive already resident in HBM (pinned, side stream)
    torch.manual_seed(1234)                           # identical weights on every rank
    model = models.setup(opt).to(dev)
    torch.manual_seed(opt.seed + 7919 * rank)         # ... and its own dropout masks / sampling noise (Philox seed = f(initial_seed))
    infos = {}
    if opt.start_from:
        model.load_state_dict(torch.load(os.path.join(opt.start_from, 'model.pth'), map_location=dev))
        infos = misc.load_infos(opt.start_from, opt.id)                          # tools/train.py:50-66
    flat = model.flatten_parameters_()
    overlap = world > 1 and os.environ.get('CAPMI_DDP_OVERLAP', '0') == '1'      # default: ONE all-reduce per step
    if overlap:
        flat.begin_overlap()
    lw_model = LossWrapper(model, opt)
    model.train()
    sched = misc.LRSchedule(opt, model_size=getattr(model, 'd_model', None))
    it, epoch = int(infos.get('iter', 0)), int(infos.get('epoch', 0))
    if opt.start_from:
        # tools/train.py:112-119: the optimizer resumes too -- Adam moments + step count (bias correction), the rate
        # schedule's state, the iteration / epoch counters (they drive warm-up, decay, ss_prob and the XE->SCST switch)
        # and the loader position
        osd = misc.load_optimizer_state(opt.start_from)
        if osd is not None and 'flat' in osd:                                    # a checkpoint of this trainer
            flat.load_state_dict(osd['flat'])
            sched.load_state_dict(osd['sched'])
        elif osd is not None:
            # a checkpoint of the REFERENCE trainer: torch.optim.Adam's state_dict (misc.py:125, tools/train.py:112-119) -> the flat
            # moment buffers; the rate schedule restarts from the iteration / epoch counters of infos
            if flat.load_torch_adam_state(osd):
                print('optimizer.pth: torch Adam state converted to the flat moment buffers (step %d)' % flat.step_count)
            else:
                print('optimizer.pth does not match this model (not a torch Adam state of its parameters): optimizer starts fresh')
        base = getattr(loader, 'loader', loader)                                 # under the prefetcher (nothing is prefetched yet)
        base = getattr(base, 'loader', base)                                     # ... and the resident feature store
        # the reference saves its sampler's index_list / iter_counter (dataloader.py:376-405): the shuffled order of the epoch
        # and both RNG streams, or a resumed epoch would revisit some images and skip others
        if hasattr(base, 'load_state'):
            for split, pos in infos.get('loader_pos', {}).items():
                base.load_state(split, order=infos.get('loader_order', {}).get(split), pos=pos)
            # only rank 0 writes infos: its caption-choice stream (seed + 104729 * rank, feature_loader.py) is rank 0's own.  The
            # other ranks keep the stream their constructor derived for THEIR rank instead of collapsing onto rank 0's
            base.load_state('train', rng=infos.get('loader_rng'), cap_rng=infos.get('loader_cap_rng') if rank == 0 else None)
        else:
            for split, pos in infos.get('loader_pos', {}).items():
                base.pos[split] = pos
        model._rng_calls = int(infos.get('rng_calls', 0))                        # dropout / sampling Philox stream position
    # r6: the iteration is graph_step.TrainStep -- forward + loss + backward + (all-reduce) + clip/Adam as one object: launch by launch
    # for UpDown / NewFC (native rollouts), captured into a hipGraph per input shape and replayed for the Transformer and AoA families
    # (450-600 launches a step from Python otherwise; CAPMI_GRAPH_STEP=0 steps everywhere).  The bucketed-overlap exchange mode keeps
    # the explicit loop below.  Built after a resume: it takes Adam's step count and the random streams' epoch from flat / the model.
    from imagecaptioning.pytorch_amd.graph_step import TrainStep
    ts = None if overlap else TrainStep(lw_model, flat, opt, dev, world=world)
    sc_ready = False
    epoch_done = True
    train_loss = float('nan')

    last_pos = {}
    last_loader = {}
    histories = dict(infos.get('histories') or {'loss_history': {}, 'lr_history': {}, 'ss_prob_history': {}})
    best_val_score = infos.get('best_val_score')

    def checkpoint():
        # the prefetcher runs ahead of the loop: the position to resume from is the one of the last CONSUMED batch
        misc.save_checkpoint(opt, model, {'iter': it, 'epoch': epoch, 'opt': opt, 'vocab': opt.vocab,
                                          'loader_pos': dict(last_pos),
                                          # (position of the dropout / sampling streams: the step record's epoch word under TrainStep)
                                          'rng_calls': ts.state_dict()['epoch'] if ts is not None else int(getattr(model, '_rng_calls', 0)),
                                          'best_val_score': best_val_score, 'histories': histories, **last_loader},
                             optimizer_state={'flat': flat.state_dict(), 'sched': sched.state_dict()})

    iter_times = []
    loss_slots = [(torch.empty(1, dtype=torch.float32, pin_memory=True), torch.cuda.Event()) for _ in range(2)]
    prev_slot = None
    while it < opt.max_iters and (opt.max_epochs == -1 or epoch < opt.max_epochs):     # tools/train.py:279-280
        if epoch_done:
            sched.epoch_start(epoch)
            if misc.scheduled_sampling_prob(opt, epoch) > 0:                 # tools/train.py:142-146
                opt.ss_prob = model.ss_prob = misc.scheduled_sampling_prob(opt, epoch)
            epoch_done = False
        sc_flag = opt.self_critical_after != -1 and epoch >= opt.self_critical_after
        struc_flag = opt.structure_after != -1 and epoch >= opt.structure_after
        if (sc_flag or struc_flag) and not sc_ready:
            # train.py:152,159 init_scorer(opt.cached_tokens): the prepro_ngrams pickle (or its converted image) when it exists,
            # else the same table rebuilt from the training references
            ct = str(opt.cached_tokens)
            have = any(os.path.exists(p_) for p_ in (ct, os.path.join('data', ct + '.p'), os.path.join('data', ct + '.capmi.npz')))
            rewards.init_scorer(ct if (opt.input_json and have) else loader.document_frequency(), device=dev)
            sc_ready = True
        t0 = time.time()
        data = loader.get_batch('train')
        fc, att, labels, masks, att_masks = (data[k] for k in ('fc_feats', 'att_feats', 'labels', 'masks', 'att_masks'))
        t1 = time.time()
        # tools/train.py:160-165, 187-191: after drop_worst_after the rows with the highest loss are left out of the mean
        drop_worst_flag = opt.drop_worst_after != -1 and epoch >= opt.drop_worst_after
        opt.current_lr = sched.rate(it)
        if ts is not None:
            loss, out = ts(data, sc_flag, struc_flag, lr=opt.current_lr, drop_worst_flag=drop_worst_flag)
        else:
            out = lw_model(fc, att, labels, masks, att_masks, data['gts'], torch.arange(len(data['gts'])), sc_flag, struc_flag,
                           drop_worst_flag)
            if not drop_worst_flag:
                loss = out['loss'].mean()
            else:
                rows = out['loss']
                loss = torch.topk(rows, k=int(rows.shape[0] * (1 - opt.drop_worst_rate)), largest=False)[0].mean()
            flat.zero_grad()
            two = struc_flag and 0 < opt.structure_loss_weight < 1                # XE + structure rollouts: two native backwards
            flat.expect_backwards(2 if two else 1)
            loss.backward()
            flat.collect_grads()
            clip = opt.grad_clip_value if opt.grad_clip_mode == 'value' else 0.0
            # buckets finished by the backward are already in flight; clip+Adam follows each as it lands
            flat.finish_overlap_and_step(opt.current_lr, (opt.optim_alpha, opt.optim_beta), opt.optim_epsilon, opt.weight_decay,
                                         clip_value=clip)
        # The reference reads the loss back right here (train.py:197), leaving the GPU idle while the host prepares the next
        # iteration.  Same values, one iteration later: the loss goes to a pinned buffer asynchronously and the host waits for the
        # PREVIOUS iteration's copy, so one iteration is always queued behind the running one (never more: the host stays at most
        # one iteration ahead).  Iterations that print, validate or checkpoint wait for their own loss.
        slot = loss_slots[it & 1]
        slot[0].copy_(loss.detach().reshape(1), non_blocking=True)
        slot[1].record()
        if prev_slot is not None and os.environ.get('CAPMI_TRAIN_LAG') != '-1':    # (-1: stress mode, the host is never throttled)
            prev_slot[1].synchronize()
            train_loss = float(prev_slot[0][0])
        wait_now = (it % opt.losses_log_every == 0 or it + 1 >= opt.max_iters or os.environ.get('CAPMI_TRAIN_LAG', '1') == '0'
                    or (opt.val_every and (it + 1) % opt.val_every == 0)
                    or (opt.save_checkpoint_every and (it + 1) % opt.save_checkpoint_every == 0))
        if wait_now:
            slot[1].synchronize()
            train_loss = float(slot[0][0])
            torch.cuda.synchronize()
        prev_slot = None if wait_now else slot
        t2 = time.time()
        iter_times.append(t2 - t0)
        if rank == 0 and it % opt.losses_log_every == 0:
            if struc_flag:
                print('iter %d (epoch %d), train_loss = %.3f, lm_loss = %.3f, struc_loss = %.3f, time/batch = %.3f'
                      % (it, epoch, train_loss, out['lm_loss'].mean().item(), out['struc_loss'].mean().item(), t2 - t1))
            elif not sc_flag:
                print('iter %d (epoch %d), train_loss = %.3f, time/batch = %.3f' % (it, epoch, train_loss, t2 - t1))
            else:
                print('iter %d (epoch %d), avg_reward = %.3f, time/batch = %.3f' % (it, epoch, out['reward'].mean().item(), t2 - t1))
            print('Read data:', t1 - t0)
        if it % opt.losses_log_every == 0:                                    # tools/train.py:214-222 histories
            histories['loss_history'][it] = train_loss
            histories['lr_history'][it] = opt.current_lr
            histories['ss_prob_history'][it] = getattr(model, 'ss_prob', 0.0)
        it += 1
        last_pos['train'] = data['bounds']['it_pos_now']
        if 'loader_state' in data['bounds']:                                  # order + RNG streams as of this batch (FeatureLoader)
            last_loader.update(data['bounds']['loader_state'])
        if data['bounds']['wrapped']:
            epoch += 1
            epoch_done = True
        if opt.val_every and it % opt.val_every == 0:
            val_sum, val_n = validation_loss(lw_model, loader, opt, dev, world)  # eval_utils.eval_split's loss half (:228-256)
            if world > 1:
                # every rank evaluates its own images: the plateau decision must see ONE number, or the ranks pick different
                # learning rates and the replicas drift apart.  (sum, count) are reduced, not per-rank means: partitions of
                # unequal size (or an empty one) must not bias the mean
                v = torch.tensor([val_sum, float(val_n)], dtype=torch.float64, device=dev)
                dist.all_reduce(v)
                val_sum, val_n = float(v[0]), float(v[1])
            val_loss = val_sum / max(val_n, 1.0)
            sched.plateau_step(val_loss)
            if best_val_score is None or -val_loss > best_val_score:           # tools/train.py:258-266 (language_eval off: -val_loss)
                best_val_score = -val_loss
            if rank == 0:
                print('validation loss: %.3f (lr %.2e)' % (val_loss, sched.current_lr))
        if rank == 0 and opt.save_checkpoint_every and it % opt.save_checkpoint_every == 0:
            checkpoint()
    if prev_slot is not None:            # the loop ended on max_epochs: the last iteration's loss is still in flight
        prev_slot[1].synchronize()
        train_loss = float(prev_slot[0][0])
    if rank == 0 and opt.save_checkpoint_every:
        checkpoint()
    if rank == 0 and len(iter_times) >= 20:        # loader + step + per-iteration host sync, second half of the run
        tail = iter_times[len(iter_times) // 2:]
        ms = 1e3 * sum(tail) / len(tail)
        print('mean time/iteration over the last %d iterations: %.2f ms = %.0f captions/s per GPU'
              % (len(tail), ms, opt.batch_size * opt.seq_per_img / ms * 1e3))
    if world > 1:
        dist.destroy_process_group()
    return train_loss


if __name__ == '__main__':
    from captioning.utils import opts
    train(opts.parse_opt())
