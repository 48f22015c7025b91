"""captioning/models/utils.py of the reference (3-24): B -> B*n row expansion helpers.

The HIP kernels never need the expanded copies (they index ``row // n``); these exist for API
parity with callers that expand tensors themselves (beam search, ensembles)."""
import torch


from imagecaptioning.pytorch_amd.ops import clip_len      # noqa: F401  (clip_att's K, cached on the mask)


def repeat_tensors(n, x):
    """[B, ...] -> [B*n, ...], rows of one image adjacent (reference: models/utils.py:3-14)."""
    if torch.is_tensor(x):
        return x.repeat_interleave(n, dim=0)
    if isinstance(x, (list, tuple)):
        return [repeat_tensors(n, v) for v in x]
    return x


def split_tensors(n, x):
    """inverse grouping (reference: models/utils.py:17-24)."""
    if torch.is_tensor(x):
        assert x.shape[0] % n == 0
        return x.reshape(x.shape[0] // n, n, *x.shape[1:]).unbind(1)
    if isinstance(x, (list, tuple)):
        return [split_tensors(n, v) for v in x]
    if x is None:
        return [None] * n
    return x


def parse_sample_method(sample_method, temperature):
    """CaptionModel.sample_next_word (CaptionModel.py:370-407) -> (kernel mode, temperature, top_k, top_p).
    'gumbel': arg-max of (logp + Gumbel) / T is a categorical draw from softmax(logp) -- the temperature cancels in the
    arg-max -- i.e. the 'sample' kernel at temperature 1.  'top<k>' keeps the k most probable tokens, 'top<p>' (0<p<1)
    the nucleus of softmax(logp / T)."""
    if sample_method == 'greedy':
        return 'greedy', temperature, 0, 0.0
    if sample_method == 'sample':
        return 'sample', temperature, 0, 0.0
    if sample_method == 'gumbel':
        return 'sample', 1.0, 0, 0.0
    if sample_method.startswith('top'):
        num = float(sample_method[3:])
        return ('sample', temperature, 0, num) if 0 < num < 1 else ('sample', temperature, int(num), 0.0)
    raise NotImplementedError('sample_method %r' % sample_method)
