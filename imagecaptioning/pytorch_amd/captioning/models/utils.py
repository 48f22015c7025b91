"""captioning/models/utils.py of the reference (3-24): B -> B*n row expansion helpers.

The HIP kernels never need the expanded copies (they index ``row // n``); these exist for API
parity with callers that expand tensors themselves (beam search, ensembles)."""
import torch


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
