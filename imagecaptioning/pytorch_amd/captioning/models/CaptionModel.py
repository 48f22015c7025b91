"""CaptionModel base: mode dispatch (reference CaptionModel.py:29-33)."""
import torch.nn as nn


class CaptionModel(nn.Module):
    def forward(self, *args, **kwargs):
        """``model(..., mode='forward'|'sample')`` -> ``self._forward`` / ``self._sample``."""
        mode = kwargs.pop('mode', 'forward')
        return getattr(self, '_' + mode)(*args, **kwargs)
