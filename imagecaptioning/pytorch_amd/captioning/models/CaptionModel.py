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

    # ---- parameters without a walk of the module tree per step.  nn.Module.named_parameters() recurses through named_modules()
    # building prefixed names and a de-duplication set: 0.5 ms for the Transformer's 260 parameters, and every forward asks twice
    # (r4: the GPU idled that long at each step's start).  The walk is cached as (name, owning module's _parameters dict, key), so
    # a Parameter OBJECT that is swapped (m.weight = nn.Parameter(...)) is still found; changes of the module TREE drop the cache
    # when they go through this module's own _apply / load_state_dict / attribute assignment / add_module / register_parameter --
    # surgery on a SUBMODULE after the first forward has to call _invalidate_param_cache().
    def _invalidate_param_cache(self):
        self.__dict__.pop('_pcache', None)

    def _param_slots(self):
        c = self.__dict__.get('_pcache')
        if c is None:
            c = []
            for name, _ in nn.Module.named_parameters(self):
                prefix, _, key = name.rpartition('.')
                c.append((name, self.get_submodule(prefix)._parameters, key))
            self.__dict__['_pcache'] = c
        return c

    def _named_param_list(self):
        """[(name, Parameter)] in named_parameters() order"""
        return [(n, d[k]) for n, d, k in self._param_slots()]

    def _param_list(self):
        return [d[k] for _, d, k in self._param_slots()]

    def _param_name_list(self):
        return [n for n, _, _ in self._param_slots()]

    def _apply(self, fn, *a, **kw):
        self._invalidate_param_cache()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._invalidate_param_cache()
        return super().load_state_dict(*a, **kw)

    def add_module(self, name, module):
        self._invalidate_param_cache()
        return super().add_module(name, module)

    def register_parameter(self, name, param):
        self._invalidate_param_cache()
        return super().register_parameter(name, param)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__.pop('_pcache', None)
        super().__setattr__(name, value)

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
