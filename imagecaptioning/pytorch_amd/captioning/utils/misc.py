"""Small host utilities of the reference's captioning/utils/misc.py that callers of the hot path use."""
import os
import pickle

import torch


# misc.py:17-18 (the post-processing list of decode_sequence; the decode-time constraint uses AttModel's own list)
DECODE_BAD_ENDINGS = frozenset(['with', 'in', 'on', 'of', 'a', 'at', 'to', 'for', 'an', 'this', 'his', 'her', 'that', 'the'])


def decode_sequence(ix_to_word, seq):
    """misc.py:62-84: token rows -> strings, stop at the first 0; with REMOVE_BAD_ENDINGS=1 in the environment the trailing
    run of function words is cut (unless the caption consists of nothing else); BPE continuation marks '@@ ' are joined."""
    strip = bool(int(os.getenv('REMOVE_BAD_ENDINGS', '0')))
    out = []
    for row in seq.tolist():
        words = []
        for ix in row:
            if ix <= 0:
                break
            words.append(ix_to_word[str(ix)])
        if strip:
            tail = 0
            while tail < len(words) and words[-1 - tail] in DECODE_BAD_ENDINGS:
                tail += 1
            if tail < len(words):
                words = words[:len(words) - tail]
        out.append(' '.join(words).replace('@@ ', ''))
    return out


def save_checkpoint(opt, model, infos, optimizer_state=None, append=''):
    """misc.py:87-102: model[-suffix].pth (state_dict), infos_{id}[-suffix].pkl."""
    suffix = ('-' + append) if append else ''
    os.makedirs(opt.checkpoint_path, exist_ok=True)
    path = os.path.join(opt.checkpoint_path, 'model%s.pth' % suffix)
    torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)
    if optimizer_state is not None:
        torch.save(optimizer_state, os.path.join(opt.checkpoint_path, 'optimizer%s.pth' % suffix))
    with open(os.path.join(opt.checkpoint_path, 'infos_%s%s.pkl' % (opt.id, suffix)), 'wb') as f:
        pickle.dump(infos, f)
    return path


def load_infos(start_from, id_):
    """tools/train.py:50-66: infos_<id>.pkl of the run being resumed ({} when absent)."""
    path = os.path.join(start_from, 'infos_%s.pkl' % id_)
    if not os.path.isfile(path):
        return {}
    with open(path, 'rb') as f:
        return pickle.load(f)


def load_optimizer_state(start_from):
    """tools/train.py:112-119: optimizer.pth next to model.pth (None when absent)."""
    path = os.path.join(start_from, 'optimizer.pth')
    return torch.load(path, map_location='cpu', weights_only=False) if os.path.isfile(path) else None


class LRSchedule:
    """Host-side learning-rate policy of the reference's training loop, for the fused clip+Adam step on the flat buffer
    (which takes the rate as a launch argument, so no torch optimizer object is involved):
      * epoch-wise exponential decay            tools/train.py:134-141  (learning_rate_decay_start/every/rate)
      * linear warm-up                          tools/train.py:171-173  (use_warmup, noamopt_warmup)
      * Noam schedule                           misc.py:159-185 NoamOpt.rate(), get_std_opt(): lr ignores learning_rate
      * reduce on plateau                       misc.py:199-216 -> torch ReduceLROnPlateau(mode='min', rel threshold 1e-4),
                                                stepped with the validation loss (tools/train.py:253-256)
    """

    def __init__(self, opt, model_size=None):
        g = lambda k, d: getattr(opt, k, d)                                   # noqa: E731
        self.base = float(opt.learning_rate)
        self.noamopt = bool(g('noamopt', False))
        self.factor = float(g('noamopt_factor', 1.0))
        self.warmup = int(g('noamopt_warmup', 2000))
        self.use_warmup = bool(g('use_warmup', False))
        self.model_size = int(model_size if model_size is not None else g('d_model', g('rnn_size', 512)))
        self.decay_start = int(g('learning_rate_decay_start', -1))
        self.decay_every = int(g('learning_rate_decay_every', 3))
        self.decay_rate = float(g('learning_rate_decay_rate', 0.8))
        self.reduce_on_plateau = bool(g('reduce_on_plateau', False))
        self.rop_factor = float(g('reduce_on_plateau_factor', 0.5))
        self.rop_patience = int(g('reduce_on_plateau_patience', 3))
        self.current_lr = self.base
        self._best, self._bad = float('inf'), 0

    def noam_rate(self, step):
        step = max(int(step), 1)
        return self.factor * (self.model_size ** -0.5) * min(step ** -0.5, step * self.warmup ** -1.5)

    def epoch_start(self, epoch):
        """tools/train.py:132-141: called when an epoch starts (and once before the first iteration)."""
        if self.noamopt or self.reduce_on_plateau:
            return self.current_lr
        if self.decay_start >= 0 and epoch > self.decay_start:
            frac = (epoch - self.decay_start) // self.decay_every
            self.current_lr = self.base * self.decay_rate ** frac
        else:
            self.current_lr = self.base
        return self.current_lr

    def rate(self, iteration):
        """learning rate of optimisation step number `iteration` (0-based, like tools/train.py's counter)."""
        if self.noamopt:
            self.current_lr = self.noam_rate(iteration + 1)                  # NoamOpt.step increments before rate()
        elif self.use_warmup and iteration < self.warmup:
            self.current_lr = self.base * (iteration + 1) / self.warmup       # tools/train.py:171-173
        return self.current_lr

    def plateau_step(self, val_loss):
        """ReduceLROnPlateau(mode='min', threshold=1e-4 rel, cooldown=0): lower the rate after `patience` evaluations
        without a relative improvement."""
        if not self.reduce_on_plateau:
            return self.current_lr
        if val_loss < self._best * (1.0 - 1e-4):
            self._best, self._bad = val_loss, 0
        else:
            self._bad += 1
        if self._bad > self.rop_patience:
            new = self.current_lr * self.rop_factor
            if self.current_lr - new > 1e-8:
                self.current_lr = new
            self._bad = 0
        return self.current_lr

    def state_dict(self):
        return dict(current_lr=self.current_lr, best=self._best, bad=self._bad)

    def load_state_dict(self, sd):
        self.current_lr, self._best, self._bad = sd['current_lr'], sd['best'], sd['bad']


def scheduled_sampling_prob(opt, epoch):
    """tools/train.py:142-146."""
    start = getattr(opt, 'scheduled_sampling_start', -1)
    if start >= 0 and epoch > start:
        frac = (epoch - start) // getattr(opt, 'scheduled_sampling_increase_every', 5)
        return min(getattr(opt, 'scheduled_sampling_increase_prob', 0.05) * frac, getattr(opt, 'scheduled_sampling_max_prob', 0.25))
    return 0.0
