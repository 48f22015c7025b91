"""Model factory (reference captioning/models/__init__.py:20-73), hot-path models only."""
from .AttModel import AttModel, UpDownModel  # noqa: F401

_OUT_OF_SCOPE = ('fc', 'show_tell', 'language_model', 'att2in', 'att2in2', 'att2all2', 'adaatt', 'adaattmo', 'stackatt',
                 'denseatt', 'bert', 'm2transformer')


def setup(opt):
    name = opt.caption_model
    if name in ('topdown', 'updown'):
        return UpDownModel(opt)
    if name == 'newfc':
        from .NewFCModel import NewFCModel
        return NewFCModel(opt)
    if name == 'transformer':
        from .TransformerModel import TransformerModel
        return TransformerModel(opt)
    if name == 'aoa':
        from .AoAModel import AoAModel
        return AoAModel(opt)
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError('caption_model %r is outside the accelerated hot path (SURVEY.md 2.1); use the '
                                  'reference implementation for it' % name)
    raise Exception('Caption model not supported: {}'.format(name))
