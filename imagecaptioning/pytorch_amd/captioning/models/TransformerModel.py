"""TransformerModel (reference captioning/models/TransformerModel.py:237-362) on the HIP backend -- BASELINE
configs[3].  The module tree below only holds parameters under the reference's names;
the arithmetic is transformer_engine.py + csrc/transformer.hip + the MFMA GEMM."""
import copy
import math

import torch
import torch.nn as nn

from .CaptionModel import CaptionModel
from imagecaptioning.pytorch_amd import transformer_engine as engine
from imagecaptioning.pytorch_amd._lib import CapmiError
from imagecaptioning.pytorch_amd.ops import clip_len


def _clones(m, n):
    return nn.ModuleList([copy.deepcopy(m) for _ in range(n)])


class _LayerNorm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(d))
        self.b_2 = nn.Parameter(torch.zeros(d))


class _Sublayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.norm = _LayerNorm(d)


class _MHA(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.linears = _clones(nn.Linear(d, d), 4)


class _FF(nn.Module):
    def __init__(self, d, dff):
        super().__init__()
        self.w_1 = nn.Linear(d, dff)
        self.w_2 = nn.Linear(dff, d)


class _EncLayer(nn.Module):
    def __init__(self, d, dff):
        super().__init__()
        self.self_attn = _MHA(d)
        self.feed_forward = _FF(d, dff)
        self.sublayer = _clones(_Sublayer(d), 2)


class _DecLayer(nn.Module):
    def __init__(self, d, dff):
        super().__init__()
        self.self_attn = _MHA(d)
        self.src_attn = _MHA(d)
        self.feed_forward = _FF(d, dff)
        self.sublayer = _clones(_Sublayer(d), 3)


class _Stack(nn.Module):
    def __init__(self, layer, n, d):
        super().__init__()
        self.layers = _clones(layer, n)
        self.norm = _LayerNorm(d)


class _Emb(nn.Module):
    def __init__(self, d, vocab):
        super().__init__()
        self.lut = nn.Embedding(vocab, d)


class _PE(nn.Module):
    def __init__(self, d, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))


class _Gen(nn.Module):
    def __init__(self, d, vocab):
        super().__init__()
        self.proj = nn.Linear(d, vocab)


class _EncDec(nn.Module):
    def __init__(self, d, dff, n_enc, n_dec, vocab):
        super().__init__()
        self.encoder = _Stack(_EncLayer(d, dff), n_enc, d)
        self.decoder = _Stack(_DecLayer(d, dff), n_dec, d)
        self.tgt_embed = nn.Sequential(_Emb(d, vocab), _PE(d))
        self.generator = _Gen(d, vocab)


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, att_feats, att_masks, seq, n, sink, seed, raw, *params):
        """seed / raw: None / False for the reference's _forward; _sample's teacher-forced gradient pass hands in the seed its
        rollout drew under and whether that rollout returned logits (explicit arguments: nothing is left on the model)."""
        P = model._pdict(params)
        ctx.sink = sink
        ctx.set_materialize_grads(False)        # the dense log-prob gradient may be undefined (sparse route)
        grads = model._grad_targets(P)
        g = engine.TransformerGraph(P, grads, model.h, model.N_enc, model.N_dec, model.drop_prob_lm, model.dropout,
                                    model.training, model._next_seed() if seed is None else seed)
        # TransformerModel._forward repeats the embedded regions BEFORE the encoder (TransformerModel.py:316-321,343-345): in train
        # mode every caption row has its own encoder dropout masks.  A rollout's gradient pass (seed given) re-runs _sample, which
        # encodes per image (:306-311).  With dropout off the rows of an image are identical copies: encode once per image.
        per_caption = (seed is None and n > 1 and model.training and model.dropout > 0
                       and not getattr(model, 'tie_encoder_dropout', False))
        g.encode(att_feats, att_masks, rows_per_image=n if per_caption else 1)
        logp = g.decode(seq, 1 if per_caption else n, raw=raw)
        ctx.g, ctx.model, ctx.grads = g, model, grads
        # (an ALIAS of the engine's tensor is returned: autograd hangs this Function on the returned object, and returning the very
        #  tensor the saved engine holds would close a reference cycle ctx -> engine -> tensor -> grad_fn -> ctx -- every activation
        #  of the step then lives until the interpreter's cyclic collector happens to run, not until the step's graph is dropped)
        return logp.detach()

    @staticmethod
    def backward(ctx, g_logp):
        flat = ctx.model._flat
        stash = flat.begin_backward() if flat is not None else None
        from imagecaptioning.pytorch_amd import sparse_logp
        g_logp, sparse, keep = sparse_logp.split_grad(g_logp, ctx.sink, ctx.g.logp)
        ctx.g._sparse_keep = keep
        ctx.g.backward(g_logp, sparse=sparse)
        if flat is not None:
            flat.end_backward(stash)
            return (None,) * (8 + len(ctx.model._param_names))
        return (None,) * 8 + tuple(ctx.grads[k] for k in ctx.model._param_names)


class TransformerModel(CaptionModel):
    graph_step = True      # graph_step.TrainStep captures this family's training iteration into a hipGraph (no host sync in it)

    def __init__(self, opt):
        super().__init__()
        self.vocab_size = opt.vocab_size
        self.seq_length = getattr(opt, 'max_length', 20) or opt.seq_length
        self.att_feat_size = opt.att_feat_size
        self.drop_prob_lm = opt.drop_prob_lm
        self.N_enc = getattr(opt, 'N_enc', opt.num_layers)
        self.N_dec = getattr(opt, 'N_dec', opt.num_layers)
        self.d_model = getattr(opt, 'd_model', opt.input_encoding_size)
        self.d_ff = getattr(opt, 'd_ff', opt.rnn_size)
        self.h = getattr(opt, 'num_att_heads', 8)
        self.dropout = getattr(opt, 'dropout', 0.1)
        self.vocab = opt.vocab
        self.ss_prob = 0.0
        if getattr(opt, 'use_bn', 0):
            raise NotImplementedError('use_bn is outside the BASELINE configs')
        self.att_embed = nn.Sequential(nn.Linear(self.att_feat_size, self.d_model), nn.ReLU(), nn.Dropout(self.drop_prob_lm))
        self.model = _EncDec(self.d_model, self.d_ff, self.N_enc, self.N_dec, self.vocab_size + 1)
        for p in self.model.parameters():              # TransformerModel.py:256-258
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        # opt-in optimisation (NOT the reference's train-mode dataflow): one encoder pass per image in _forward even when dropout is
        # on -- the n caption rows of an image then share their encoder dropout masks (same expected gradient, correlated noise)
        import os
        self.tie_encoder_dropout = bool(getattr(opt, 'tie_encoder_dropout', 0)) or os.environ.get('CAPMI_TIE_ENC') == '1'
        self._flat = None
        self._rng_calls = 0

    # ---- plumbing
    @property
    def _param_names(self):
        return self._param_name_list()

    def _pdict(self, params):
        P = dict(zip(self._param_names, [p.detach() for p in params]))
        P['model.tgt_embed.1.pe'] = self.model.tgt_embed[1].pe[0].contiguous()      # [max_len, D]
        return P

    def _grad_targets(self, P):
        if self._flat is not None:
            return self._flat.grad_views
        return {k: torch.empty_like(v) for k, v in P.items() if k != 'model.tgt_embed.1.pe'}

    def _flat_groups(self):
        """parameters the engine wants back to back in the flat buffers (transformer_engine.fused_lin): Wq | Wk | Wv (and their
        biases) of every self-attention block, and Wk | Wv of the cross-attention of ALL decoder layers"""
        names = [n for n, _ in self.named_parameters()]
        blocks = sorted({n[:n.index('.self_attn.') + len('.self_attn')] for n in names if '.self_attn.linears.' in n})
        groups = []
        for b in blocks:
            for kind in ('weight', 'bias'):
                groups.append(['%s.linears.%d.%s' % (b, i, kind) for i in range(3)])
        n_dec = len({n.split('.')[3] for n in names if n.startswith('model.decoder.layers.')})
        for kind in ('weight', 'bias'):
            groups.append(['model.decoder.layers.%d.src_attn.linears.%d.%s' % (i, j, kind) for i in range(n_dec) for j in (1, 2)])
        return groups

    def flatten_parameters_(self):
        from imagecaptioning.pytorch_amd.flat import FlatParams
        self._flat = FlatParams(self)
        return self._flat

    def _next_seed(self):
        self._rng_calls += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._rng_calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def logit(self, x):
        return torch.nn.functional.linear(x, self.model.generator.proj.weight, self.model.generator.proj.bias)

    def init_hidden(self, bsz):
        return []

    def _clip(self, att_feats, att_masks):
        if att_masks is not None:
            ml = clip_len(att_masks)
            att_feats, att_masks = att_feats[:, :ml].contiguous(), att_masks[:, :ml].contiguous().float()
        return att_feats.float().contiguous(), att_masks

    # ---- reference API
    def _forward(self, fc_feats, att_feats, seq, att_masks=None, _seed=None, _raw=False):
        """TransformerModel._forward (:340-348): log-probs [N,T,V1]."""
        if not att_feats.is_cuda:
            raise CapmiError('the capmi backend runs on a HIP device only; there is no CPU path')
        if seq.ndim == 3:
            seq = seq.reshape(-1, seq.shape[2])
        seq = seq.long().contiguous()
        att_feats, att_masks = self._clip(att_feats, att_masks)
        n = seq.shape[0] // att_feats.shape[0]
        params = self._param_list()
        from imagecaptioning.pytorch_amd import sparse_logp
        sink = sparse_logp.LogpSink()
        return sparse_logp.attach(_Fn.apply(self, att_feats, att_masks, seq, n, sink, _seed, bool(_raw), *params), sink)

    def _sample(self, fc_feats, att_feats, att_masks=None, opt={}):
        """AttModel._sample for the Transformer.  Tokens are drawn with the KV-cached decoder under no_grad; when a
        gradient is needed (SCST) the log-probs of the drawn tokens are recomputed by ONE teacher-forced pass.  In train
        mode both run under ONE dropout realisation (same Philox seed; step t of the rollout applies position t's rows of the
        teacher-forced pass's masks), so the differentiated pass is the sampled pass, as in the reference, which backpropagates
        through the very steps it sampled from (loss_wrapper.py:63-68).  (The reference re-decodes the whole prefix at every
        step, TransformerModel.py:351-362, and so redraws the masks of earlier positions each step; here a position keeps its
        masks for the whole rollout -- the KV cache's meaning -- which is the same policy-gradient estimator for a slightly
        different, equally valid, noise model.)"""
        method = opt.get('sample_method', 'greedy')
        from imagecaptioning.pytorch_amd import decode
        raw = not opt.get('output_logsoftmax', 1)
        if raw and ((opt.get('beam_size', 1) > 1 and method in ('greedy', 'beam_search')) or decode.wants_options(opt)):
            # AttModel.py:171-175: the margin structure losses read raw LOGITS (loss_wrapper.py:31-37 samples them with sample_n and no
            # decode-time option); the sampled / greedy rollout and its teacher-forced gradient pass return them (r5), beam search
            # and the option samplers return log-probabilities -- refuse rather than hand those to a margin loss
            raise NotImplementedError('output_logsoftmax=0 is implemented for the sampled / greedy rollout; beam search and the '
                                      'decode-time options of %s return log-probabilities' % type(self).__name__)
        if not att_feats.is_cuda:
            raise CapmiError('the capmi backend runs on a HIP device only; there is no CPU path')
        if opt.get('beam_size', 1) > 1 and method in ('greedy', 'beam_search'):
            att_feats, att_masks = self._clip(att_feats, att_masks)
            with torch.no_grad():
                P = self._pdict(self._param_list())
                return engine.sample_beam(self, P, att_feats, att_masks, self.h, self.N_enc, self.N_dec, self.seq_length, opt)
        from .utils import parse_sample_method
        if decode.wants_options(opt):
            att_feats, att_masks = self._clip(att_feats, att_masks)
            P = self._pdict(self._param_list())
            return self._sample_with_options(
                lambda rows: engine.Decoder(P, att_feats, att_masks, self.h, self.N_enc, self.N_dec, self.seq_length, rows),
                att_feats.size(0), opt)
        mode, temperature, top_k, top_p = parse_sample_method(method, opt.get('temperature', 1.0))
        n = int(opt.get('sample_n', 1))
        att_feats, att_masks = self._clip(att_feats, att_masks)
        want_grad = torch.is_grad_enabled() and self.training
        drop, drop_seed = None, None
        if want_grad and (self.dropout > 0 or self.drop_prob_lm > 0):
            drop_seed = self._next_seed()
            drop = (self.drop_prob_lm, self.dropout, drop_seed)
        with torch.no_grad():
            P = self._pdict(self._param_list())
            if mode == 'greedy' and opt.get('_graph', True) and not self.training:
                # deterministic and launch-bound on the host (~1300 launches): replay a captured hipGraph
                if not hasattr(self, '_graphs'):
                    from imagecaptioning.pytorch_amd.graphs import GraphedDecode
                    self._graphs = GraphedDecode()
                seq, logp = self._graphs(('greedy', n, self.seq_length, raw),
                                         lambda a, m: engine.sample(P, a, m, self.h, self.N_enc, self.N_dec, self.seq_length,
                                                                    sample_n=n, mode='greedy', raw=raw),
                                         (att_feats.contiguous(), att_masks))
            else:
                seq, logp = engine.sample(P, att_feats, att_masks, self.h, self.N_enc, self.N_dec, self.seq_length, sample_n=n,
                                          mode=mode, temperature=temperature, seed=self._next_seed(),
                                          gumbel=opt.get('_gumbel'), top_k=top_k, top_p=top_p, drop=drop, raw=raw)
        if not want_grad:
            return seq, logp
        # differentiable log-probs of the drawn tokens: inputs [bos, w_0 .. w_{L-2}]; the teacher-forced pass draws the rollout's
        # masks again (its seed) and returns logits when the rollout did
        inp = torch.cat([seq.new_zeros(seq.shape[0], 1), seq[:, :-1]], 1)
        logp_g = self._forward(None, att_feats, inp, att_masks, _seed=drop_seed if drop_seed is not None else self._next_seed(),
                               _raw=raw)
        live = torch.cat([seq.new_ones(seq.shape[0], 1), (seq[:, :-1] > 0).long()], 1).cumprod(1)     # unfinished-before-step
        from imagecaptioning.pytorch_amd import sparse_logp
        return seq, sparse_logp.attach_masked(logp_g * live.unsqueeze(-1).to(logp_g), logp_g, live)
