"""AoAModel (reference captioning/models/AoAModel.py:188-226) on the HIP backend -- BASELINE configs[4]
(configs/aoa.yml switches: refine 1, refine_aoa 1, use_ff 0, decoder_type AoA, use_multi_head 2, mean_feats 1).
Parameter tree = the reference's; arithmetic = aoa_engine.py + csrc kernels."""
import copy

import torch
import torch.nn as nn

from .CaptionModel import CaptionModel
from .TransformerModel import _LayerNorm, _Sublayer, _clones
from imagecaptioning.pytorch_amd import aoa_engine as engine
from imagecaptioning.pytorch_amd._lib import CapmiError
from imagecaptioning.pytorch_amd.ops import clip_len


class _MHDot(nn.Module):
    """MultiHeadedDotAttention parameter holder (AoAModel.py:17-55)."""

    def __init__(self, d, project_k_v, do_aoa, norm_q):
        super().__init__()
        if norm_q:
            self.norm = _LayerNorm(d)
        self.linears = _clones(nn.Linear(d, d), 1 + 2 * project_k_v)
        if do_aoa:
            self.aoa_layer = nn.Sequential(nn.Linear(2 * d, 2 * d), nn.GLU())


class _RefLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self_attn = _MHDot(d, 1, 1, 0)
        self.sublayer = _clones(_Sublayer(d), 1)


class _Refiner(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.layers = _clones(_RefLayer(d), 6)
        self.norm = _LayerNorm(d)


class _Core(nn.Module):
    def __init__(self, opt):
        super().__init__()
        R = opt.rnn_size
        self.att_lstm = nn.LSTMCell(opt.input_encoding_size + R, R)
        self.att2ctx = nn.Sequential(nn.Linear(2 * R, 2 * R), nn.GLU())
        self.attention = _MHDot(R, 0, 0, 1)


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, cfg, att_feats, att_masks, *params):
        P = dict(zip(model._param_names, [p.detach() for p in params]))
        ctx.sink = cfg.pop('_sink', None)
        ctx.set_materialize_grads(False)        # the dense log-prob gradient may be undefined (sparse route)
        grads = model._flat.grad_views if model._flat is not None else {k: torch.empty_like(v) for k, v in P.items()}
        g = engine.AoAGraph(P, grads, model.num_heads, model.drop_prob_lm, model.dropout_aoa, model.training, model._next_seed())
        g.prepare(att_feats, att_masks)
        seq, logp = g.rollout(**cfg)
        ctx.g, ctx.model, ctx.grads = g, model, grads
        ctx.mark_non_differentiable(seq)
        # (an ALIAS of the engine's tensor is returned: autograd hangs this Function on the returned object, and returning the very
        #  tensor the saved engine holds would close a reference cycle ctx -> engine -> tensor -> grad_fn -> ctx -- every activation
        #  of the step then lives until the interpreter's cyclic collector happens to run, not until the step's graph is dropped)
        return seq, logp.detach()

    @staticmethod
    def backward(ctx, _gs, g_logp):
        flat = ctx.model._flat
        stash = flat.begin_backward() if flat is not None else None
        from imagecaptioning.pytorch_amd import sparse_logp
        g_logp, sparse, keep = sparse_logp.split_grad(g_logp, ctx.sink, ctx.g.seq_logp)
        ctx.g._sparse_keep = keep
        ctx.g.backward(g_logp, sparse=sparse)
        if flat is not None:
            flat.end_backward(stash)
            return (None,) * (4 + len(ctx.model._param_names))
        return (None, None, None, None) + tuple(ctx.grads[k] for k in ctx.model._param_names)


class AoAModel(CaptionModel):
    graph_step = True      # graph_step.TrainStep captures this family's training iteration into a hipGraph (no host sync in it)

    def __init__(self, opt):
        super().__init__()
        for k, want in (('refine', 1), ('refine_aoa', 1), ('use_ff', 0), ('use_multi_head', 2), ('multi_head_scale', 1)):
            if getattr(opt, k, want) != want:
                raise NotImplementedError('AoA option %s=%r is outside configs/aoa.yml' % (k, getattr(opt, k)))
        if getattr(opt, 'decoder_type', 'AoA') != 'AoA' or not getattr(opt, 'mean_feats', 1):
            raise NotImplementedError('only decoder_type AoA with mean_feats 1 (configs/aoa.yml) is accelerated')
        if not getattr(opt, 'ctx_drop', 0):
            pass       # ctx_drop 0 == identity mask: handled by drop_prob in eval; train-mode ctx_drop=0 not in aoa.yml
        self.vocab_size = opt.vocab_size
        self.rnn_size = opt.rnn_size
        self.input_encoding_size = opt.input_encoding_size
        self.num_layers = 2
        self.num_heads = opt.num_heads
        self.drop_prob_lm = opt.drop_prob_lm
        self.dropout_aoa = getattr(opt, 'dropout_aoa', 0.3)
        self.seq_length = getattr(opt, 'max_length', 20) or opt.seq_length
        self.vocab = opt.vocab
        self.ss_prob = 0.0
        R = self.rnn_size
        self.embed = nn.Sequential(nn.Embedding(self.vocab_size + 1, self.input_encoding_size), nn.ReLU(),
                                   nn.Dropout(self.drop_prob_lm))
        self.att_embed = nn.Sequential(nn.Linear(opt.att_feat_size, R), nn.ReLU(), nn.Dropout(self.drop_prob_lm))
        self.logit = nn.Linear(R, self.vocab_size + 1)
        self.ctx2att = nn.Linear(R, 2 * R)
        self.refiner = _Refiner(R)
        self.core = _Core(opt)
        self._flat = None
        self._rng_calls = 0

    @property
    def _param_names(self):
        return self._param_name_list()

    def _flat_groups(self):
        """Wq | Wk | Wv (and biases) of every refiner layer back to back in the flat buffers: one fused projection GEMM each"""
        names = [n for n, _ in self.named_parameters()]
        blocks = sorted({n[:n.index('.self_attn.') + len('.self_attn')] for n in names if n.startswith('refiner.') and '.self_attn.linears.' in n})
        return [['%s.linears.%d.%s' % (b, i, kind) for i in range(3)] for b in blocks for kind in ('weight', 'bias')]

    def flatten_parameters_(self):
        from imagecaptioning.pytorch_amd.flat import FlatParams
        self._flat = FlatParams(self)
        return self._flat

    def _next_seed(self):
        self._rng_calls += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._rng_calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def _run(self, cfg, att_feats, att_masks, clipped=False):
        if not att_feats.is_cuda:
            raise CapmiError('the capmi backend runs on a HIP device only; there is no CPU path')
        if att_masks is not None and not clipped:
            ml = clip_len(att_masks)
            att_feats, att_masks = att_feats[:, :ml], att_masks[:, :ml].float().contiguous()
        params = self._param_list()
        from imagecaptioning.pytorch_amd import sparse_logp
        cfg = dict(cfg)
        cfg['_sink'] = sink = sparse_logp.LogpSink()
        seq, logp = _Fn.apply(self, cfg, att_feats.float().contiguous(), att_masks, *params)
        return seq, sparse_logp.attach(logp, sink)

    def _forward(self, fc_feats, att_feats, seq, att_masks=None):
        if self.training and self.ss_prob > 0:
            raise NotImplementedError('scheduled sampling (ss_prob > 0) is only wired into the UpDown rollout')
        B = att_feats.size(0)
        if seq.ndim == 3:
            seq = seq.reshape(-1, seq.shape[2])
        seq = seq.long().contiguous()
        N, T = seq.shape
        zero_cols = (seq[:, 1:].sum(0) == 0).nonzero()
        T_eff = int(zero_cols[0]) + 1 if zero_cols.numel() else T
        _, logp = self._run(dict(n=N // B, T=T_eff, L=T, forced=seq, teacher=True), att_feats, att_masks)
        return logp

    def _sample(self, fc_feats, att_feats, att_masks=None, opt={}):
        method = opt.get('sample_method', 'greedy')
        from imagecaptioning.pytorch_amd import decode
        raw = not opt.get('output_logsoftmax', 1)
        if raw and ((opt.get('beam_size', 1) > 1 and method in ('greedy', 'beam_search')) or decode.wants_options(opt)):
            # AttModel.py:171-175: the margin structure losses read raw LOGITS (loss_wrapper.py:31-37 samples them with sample_n and no
            # decode-time option); the sampled / greedy rollout stores them (r5, CAPMI_SELECT_RAW), beam search and the option
            # samplers return log-probabilities -- refuse rather than hand those to a margin loss
            raise NotImplementedError('output_logsoftmax=0 is implemented for the sampled / greedy rollout; beam search and the '
                                      'decode-time options of %s return log-probabilities' % type(self).__name__)
        if opt.get('beam_size', 1) > 1 and method in ('greedy', 'beam_search'):
            if not att_feats.is_cuda:
                raise CapmiError('the capmi backend runs on a HIP device only; there is no CPU path')
            if att_masks is not None:
                ml = clip_len(att_masks)
                att_feats, att_masks = att_feats[:, :ml], att_masks[:, :ml].float().contiguous()
            with torch.no_grad():
                P = dict(zip(self._param_names, [p.detach() for p in self._param_list()]))
                return engine.sample_beam(self, P, att_feats.float().contiguous(), att_masks, self.num_heads, self.seq_length, opt)
        from .utils import parse_sample_method
        if decode.wants_options(opt):
            if not att_feats.is_cuda:
                raise CapmiError('the capmi backend runs on a HIP device only; there is no CPU path')
            if att_masks is not None:
                ml = clip_len(att_masks)
                att_feats, att_masks = att_feats[:, :ml], att_masks[:, :ml].float().contiguous()
            P = dict(zip(self._param_names, [p.detach() for p in self._param_list()]))

            def make(rows):
                g = engine.AoAGraph(P, {}, self.num_heads, 0.0, 0.0, False, 0)
                g.prepare(att_feats.float().contiguous(), att_masks)
                return engine.BeamDecoder(g, rows)
            return self._sample_with_options(make, att_feats.size(0), opt)
        mode, temperature, top_k, top_p = parse_sample_method(method, opt.get('temperature', 1.0))
        L = self.seq_length
        cfg = dict(n=int(opt.get('sample_n', 1)), T=L, L=L, mode=mode, temperature=temperature,
                   seed=self._next_seed(), gumbel=opt.get('_gumbel'), top_k=top_k, top_p=top_p)
        if raw:
            cfg['raw'] = True
        if mode == 'greedy' and not self.training and not torch.is_grad_enabled() and opt.get('_graph', True) and att_feats.is_cuda:
            # deterministic, no gradient, launch-bound on the host: replay a captured hipGraph (graphs.py)
            if not hasattr(self, '_graphs'):
                from imagecaptioning.pytorch_amd.graphs import GraphedDecode
                self._graphs = GraphedDecode()
            gcfg = dict(cfg, seed=0)
            if att_masks is not None:                 # the data-dependent clip (a host sync) stays outside the graph
                ml = clip_len(att_masks)
                att_feats, att_masks = att_feats[:, :ml], att_masks[:, :ml].float().contiguous()
            return self._graphs(('greedy', cfg['n'], L, raw), lambda a, m: self._run(gcfg, a, m, clipped=True),
                                (att_feats.float().contiguous(), att_masks))
        return self._run(cfg, att_feats, att_masks)
