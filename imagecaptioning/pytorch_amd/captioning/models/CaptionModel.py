"""CaptionModel base: mode dispatch (reference CaptionModel.py:29-33) and the decode-option plumbing every family shares."""
import torch
import torch.nn as nn

BAD_ENDINGS = ['a', 'an', 'the', 'in', 'for', 'at', 'of', 'with', 'before', 'after', 'on', 'upon', 'near', 'to', 'is',
               'are', 'am', 'the']       # AttModel.py:27


class CaptionModel(nn.Module):
    def forward(self, *args, **kwargs):
        """``model(..., mode='forward'|'sample')`` -> ``self._forward`` / ``self._sample``."""
        mode = kwargs.pop('mode', 'forward')
        return getattr(self, '_' + mode)(*args, **kwargs)

    @property
    def bad_endings_ix(self):
        """AttModel.py:96-97 (every family derives from AttModel there); assignable for tests."""
        ix = self.__dict__.get('_bad_endings_ix')
        if ix is None:
            ix = [int(k) for k, v in self.vocab.items() if v in BAD_ENDINGS]
        return ix

    @bad_endings_ix.setter
    def bad_endings_ix(self, ix):
        self.__dict__['_bad_endings_ix'] = [int(i) for i in ix]

    def _sample_with_options(self, make_stepper, B, opt):
        """AttModel._sample's option branches that need per-step hooks (AttModel.py:270-271 _diverse_sample,
        :293-330 constraints): host-stepped on the family's single-step decoder.  Eval numerics, no gradient.
        make_stepper(rows_per_image) -> stepper (imagecaptioning.pytorch_amd.step protocol)."""
        from imagecaptioning.pytorch_amd import decode
        dev = next(self.parameters()).device
        with torch.no_grad():
            if opt.get('group_size', 1) > 1:
                st = make_stepper(int(opt['group_size']))
                return decode.diverse_sample_steps(self, st, B, self.seq_length, opt, dev, seed=self._next_seed())
            st = make_stepper(int(opt.get('sample_n', 1)))
            return decode.sample_steps(self, st, B, self.seq_length, opt, dev, seed=self._next_seed())
