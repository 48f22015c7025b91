"""NewFCModel (reference AttModel.py:904-945 over FCModel.LSTMCore 13-42) on the HIP backend -- BASELINE
configs[0] (configs/fc.yml).  Same parameter names as the reference: embed.weight, fc_embed.{weight,bias},
_core.i2h/h2h.{weight,bias}, logit.{weight,bias}."""
import torch
import torch.nn as nn

from .CaptionModel import CaptionModel
from imagecaptioning.pytorch_amd import newfc_engine as engine
from imagecaptioning.pytorch_amd import ops
from imagecaptioning.pytorch_amd import sparse_logp
from imagecaptioning.pytorch_amd._lib import CapmiError


class LSTMCore(nn.Module):
    """Parameter holder for FCModel.LSTMCore (FCModel.py:13-23)."""

    def __init__(self, opt):
        super().__init__()
        self.i2h = nn.Linear(opt.input_encoding_size, 5 * opt.rnn_size)
        self.h2h = nn.Linear(opt.rnn_size, 5 * opt.rnn_size)


class _RolloutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, cfg, fc_feats, *params):
        P = dict(zip(model._param_names, [p.detach() for p in params]))
        ctx.sink = cfg.pop('_sink', None)
        ctx.set_materialize_grads(False)        # the dense log-prob gradient may be undefined (sparse route)
        ro = engine.Rollout(P, fc_feats, **cfg)
        seq, logp = ro.run()
        ctx.model, ctx.ro, ctx.P = model, ro, P
        ctx.mark_non_differentiable(seq)
        # (an ALIAS of the engine's tensor is returned: autograd hangs this Function on the returned object, and returning the very
        #  tensor the saved engine holds would close a reference cycle ctx -> engine -> tensor -> grad_fn -> ctx -- every activation
        #  of the step then lives until the interpreter's cyclic collector happens to run, not until the step's graph is dropped)
        return seq, logp.detach()

    @staticmethod
    def backward(ctx, _g, g_logp):
        model, ro, P = ctx.model, ctx.ro, ctx.P
        flat = model._flat
        stash = flat.begin_backward() if flat is not None else None
        grads = flat.grad_views if flat is not None else {k: torch.empty_like(v) for k, v in P.items()}
        g_logp, sparse, keep = sparse_logp.split_grad(g_logp, ctx.sink, ro.seq_logp)
        ro._sparse_keep = keep
        ro.backward(g_logp, grads, sparse=sparse)
        if flat is not None:
            flat.end_backward(stash)
            return (None,) * (3 + len(model._param_names))
        return (None, None, None) + tuple(grads[k] for k in model._param_names)


class NewFCModel(CaptionModel):
    def __init__(self, opt):
        super().__init__()
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.num_layers = 1
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = getattr(opt, 'max_length', 20) or opt.seq_length
        self.fc_feat_size = opt.fc_feat_size
        self.ss_prob = 0.0
        self.vocab = opt.vocab
        self.fc_embed = nn.Linear(self.fc_feat_size, self.input_encoding_size)
        self.embed = nn.Embedding(self.vocab_size + 1, self.input_encoding_size)
        self._core = LSTMCore(opt)
        self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)
        self._flat = None
        self._rng_calls = 0

    @property
    def _param_names(self):
        return self._param_name_list()

    def flatten_parameters_(self):
        from imagecaptioning.pytorch_amd.flat import FlatParams
        self._flat = FlatParams(self)
        return self._flat

    def _next_seed(self):
        self._rng_calls += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._rng_calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def _run(self, cfg, fc_feats):
        if not fc_feats.is_cuda:
            raise CapmiError('the capmi backend runs on a HIP device only; there is no CPU path')
        N, T = fc_feats.shape[0] * cfg['n'], cfg['T']
        if self.training and self.drop_prob_lm > 0:
            cfg['drop_out'] = ops.dropout_mask((T, N, self.rnn_size), self.drop_prob_lm, self._next_seed(), 0,
                                               fc_feats.device)
        params = self._param_list()
        cfg['_sink'] = sink = sparse_logp.LogpSink()
        seq, logp = _RolloutFn.apply(self, cfg, fc_feats.float().contiguous(), *params)
        return seq, sparse_logp.attach(logp, sink)

    def _forward(self, fc_feats, att_feats, seq, att_masks=None):
        if self.training and self.ss_prob > 0:
            raise NotImplementedError('scheduled sampling (ss_prob > 0) is only wired into the UpDown rollout')
        B = fc_feats.size(0)
        if seq.ndim == 3:
            seq = seq.reshape(-1, seq.shape[2])
        seq = seq.long().contiguous()
        N, T = seq.shape
        zero_cols = (seq[:, 1:].sum(0) == 0).nonzero()
        T_eff = int(zero_cols[0]) + 1 if zero_cols.numel() else T
        _, logp = self._run(dict(n=N // B, T=T_eff, L=T, mode='forced', forced=seq, teacher=True), fc_feats)
        return logp

    def _sample(self, fc_feats, att_feats, att_masks=None, opt={}):
        method = opt.get('sample_method', 'greedy')
        from .utils import parse_sample_method
        from imagecaptioning.pytorch_amd import decode, beam
        from imagecaptioning.pytorch_amd.step import NewFCStepper
        raw = not opt.get('output_logsoftmax', 1)
        is_beam = opt.get('beam_size', 1) > 1 and method in ('greedy', 'beam_search')
        mode, temperature, top_k, top_p = (None, 1.0, 0, 0.0) if is_beam else parse_sample_method(method, opt.get('temperature', 1.0))
        if raw and (is_beam or decode.wants_options(opt) or top_k or top_p):
            # AttModel.py:171-175: the margin structure losses read raw LOGITS (loss_wrapper.py:31-37 samples them with sample_n and no
            # decode-time option); the one-call rollout stores them (r5, CAPMI_SELECT_RAW), beam search and the host-stepped option
            # samplers return log-probabilities -- refuse rather than hand those to a margin loss
            raise NotImplementedError('output_logsoftmax=0 is implemented for the sampled / greedy rollout; beam search and the '
                                      'decode-time options of %s return log-probabilities' % type(self).__name__)
        if not fc_feats.is_cuda:
            raise CapmiError('the capmi backend runs on a HIP device only; there is no CPU path')
        P = dict(zip(self._param_names, [p.detach() for p in self._param_list()]))
        if is_beam:
            # AttModel._sample_beam on the single-step decoder (the image step is taken once per image, AttModel.py:925-927)
            with torch.no_grad():
                return beam.beam_search_steps(self, lambda rows: NewFCStepper(P, fc_feats, rows), fc_feats.size(0),
                                              P['embed.weight'].shape[0], self.seq_length, opt, fc_feats.device)
        if decode.wants_options(opt) or top_k or top_p:
            # options the one-call rollout has no hooks for: host-stepped (eval numerics, no gradient)
            return self._sample_with_options(lambda rows: NewFCStepper(P, fc_feats, rows), fc_feats.size(0), opt)
        cfg = dict(n=int(opt.get('sample_n', 1)), T=self.seq_length, L=self.seq_length, mode=mode,
                   temperature=temperature, seed=self._next_seed())
        if raw:
            cfg['raw'] = True
        if opt.get('_gumbel') is not None:         # test hook: injected noise [L, N, V1]
            cfg['gumbel'] = opt['_gumbel']
        return self._run(cfg, fc_feats)
