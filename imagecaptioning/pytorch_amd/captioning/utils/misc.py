"""Small host utilities of the reference's captioning/utils/misc.py that callers of the hot path use."""
import os
import pickle

import torch


def decode_sequence(ix_to_word, seq):
    """misc.py:62-84: token rows -> strings, stop at the first 0."""
    out = []
    for row in seq.tolist():
        words = []
        for ix in row:
            if ix <= 0:
                break
            words.append(ix_to_word[str(ix)])
        out.append(' '.join(words))
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
