"""AttModel / UpDownModel on the MI355X HIP backend.

Drop-in for the reference's captioning/models/AttModel.py for the hot path: same constructor
(``opt`` Namespace), same parameter tree (=> same ``state_dict`` keys/shapes),
same call surface:

    model(fc_feats, att_feats, seq, att_masks)                       -> logprobs [N,T,V1]   (_forward, 126-164)
    model(fc_feats, att_feats, att_masks, opt={...}, mode='sample')  -> (seq [N,L], seqLogprobs [N,L,V1])  (_sample, 258-352)
    model.get_logprobs_state(it, fc, att, p_att, masks, state)       -> (logprobs, state)   (166-176)
    model._prepare_feature / init_hidden / embed / core / logit attributes

The nn.Module tree below only *holds parameters* (and gives the reference's default initialisation);
all arithmetic runs in libcapmi: one native call per rollout, no per-step host sync, no
repeat_tensors copies, BPTT by hand-written kernels behind one autograd.Function.
There is no CPU path: tensors must live on a HIP device.
"""
import torch
import torch.nn as nn

from .CaptionModel import CaptionModel
from imagecaptioning.pytorch_amd import updown_engine as engine
from imagecaptioning.pytorch_amd import ops
from imagecaptioning.pytorch_amd import sparse_logp
from imagecaptioning.pytorch_amd._lib import CapmiError

from .utils import parse_sample_method, clip_len      # noqa: E402


class _RolloutFn(torch.autograd.Function):
    """(params..., feats) -> (seq, dense seqLogprobs); backward = hand-written BPTT + prefill backward.
    Gradients are written into the model's flat gradient views when it has them."""

    @staticmethod
    def forward(ctx, model, cfg, fc_feats, att_feats, att_masks, *params):
        P = dict(zip(model._param_names, [p.detach() for p in params]))
        if cfg.get('fused_greedy'):
            # fused SCST rollout: sampled rows read the train-mode (dropout) features, the greedy-baseline rows the
            # eval-mode features of the same images -> 2B feature images written side by side (no torch.cat), explicit
            # row -> image map
            B = fc_feats.shape[0]
            K = clip_len(att_masks, att_feats.shape[1])
            R, A = P['fc_embed.0.weight'].shape[0], P['ctx2att.weight'].shape[0]
            dev = fc_feats.device
            pr_run = engine.Prepared()
            pr_run.fc = torch.empty(2 * B, R, dtype=torch.float32, device=dev)
            pr_run.att = torch.empty(2 * B, K, R, dtype=torch.float32, device=dev)
            pr_run.p_att = torch.empty(2 * B, K, A, dtype=torch.float32, device=dev)
            pr = engine.prepare(P, fc_feats, att_feats, att_masks, cfg.get('drop_fc'), cfg.get('drop_att'),
                                out=(pr_run.fc[:B], pr_run.att[:B], pr_run.p_att[:B]))
            engine.prepare(P, fc_feats, att_feats, att_masks, out=(pr_run.fc[B:], pr_run.att[B:], pr_run.p_att[B:]))
            pr_run.att_masks = None if pr.att_masks is None else torch.cat([pr.att_masks, pr.att_masks], 0)
            extra = dict(row_img=cfg['row_img'], B_grad=B)
        else:
            pr = engine.prepare(P, fc_feats, att_feats, att_masks, cfg.get('drop_fc'), cfg.get('drop_att'))
            pr_run, extra = pr, {}
        ro = engine.Rollout(P, pr_run, n=cfg['n'], T=cfg['T'], L=cfg['L'], mode=cfg['mode'],
                            temperature=cfg.get('temperature', 1.0), drop_xt=cfg.get('drop_xt'),
                            drop_out=cfg.get('drop_out'), gumbel=cfg.get('gumbel'), seed=cfg.get('seed', 0),
                            forced=cfg.get('forced'), teacher=cfg.get('teacher', False), row_mode=cfg.get('row_mode'),
                            top_k=cfg.get('top_k', 0), top_p=cfg.get('top_p', 0.0), ss_mode=cfg.get('ss_mode'),
                            raw_logits=cfg.get('raw_logits', False), **extra)
        seq, seq_logp = ro.run()
        ctx.model, ctx.ro, ctx.pr, ctx.P = model, ro, pr, P
        ctx.sink = sink = cfg.get('_sink')
        if sink is not None:
            sink.sel, sink.seq = ro.sel_logp, ro.seq        # what a fused criterion needs: selected log-probs + tokens
        ctx.set_materialize_grads(False)        # the dense log-prob gradient may be undefined (sparse route)
        ctx.mark_non_differentiable(seq)
        model._last_rollout = ro
        # (an alias: returning the tensor the saved rollout holds would close a reference cycle ctx -> rollout -> tensor -> grad_fn ->
        #  ctx, and the step's activations would wait for the cyclic collector; see TransformerModel._Fn)
        return seq, seq_logp.detach()

    @staticmethod
    def backward(ctx, _g_seq, g_logp):
        model, ro, pr, P = ctx.model, ctx.ro, ctx.pr, ctx.P
        flat = model._flat
        stash = flat.begin_backward() if flat is not None else None
        grads = model._grad_targets(P)
        g_logp, sparse, keep = sparse_logp.split_grad(g_logp, ctx.sink, ro.seq_logp)       # the criteria hand their gradient over sparse
        ro._sparse_keep = keep
        d_fc, d_att, d_p_att = ro.backward(g_logp, grads, sparse=sparse, on_ready=flat.on_grads_ready if (flat is not None and flat.overlap_allowed(stash)) else None)
        engine.prepare_backward(P, pr, d_fc, d_att, d_p_att, grads)
        if flat is not None:
            flat.end_backward(stash)
            return (None,) * (5 + len(model._param_names))
        return (None, None, None, None, None) + tuple(grads[k] for k in model._param_names)


class Attention(nn.Module):
    """Parameter holder for Attention (AttModel.py:719-726): h2att, alpha_net."""

    def __init__(self, opt):
        super().__init__()
        self.h2att = nn.Linear(opt.rnn_size, opt.att_hid_size)
        self.alpha_net = nn.Linear(opt.att_hid_size, 1)


class UpDownCore(nn.Module):
    """UpDownCore (AttModel.py:615-640): att_lstm, lang_lstm, attention.  The rollouts never call this module (they run in
    libcapmi from its parameters); ``forward`` is the reference's step contract for callers that drive the core themselves
    (AttEnsemble.py:52-53 ``m.core(xt, fc, att, p_att, state, masks)``), served by the same kernels: two 4-/3-segment gate
    GEMMs read the concatenated inputs in place, the LSTM cells finish their split-K slabs, fused region attention.  No
    autograd through it (inference API)."""

    def __init__(self, opt):
        super().__init__()
        self.att_lstm = nn.LSTMCell(opt.input_encoding_size + opt.rnn_size * 2, opt.rnn_size)
        self.lang_lstm = nn.LSTMCell(opt.rnn_size * 2, opt.rnn_size)
        self.attention = Attention(opt)
        self.drop_prob_lm = opt.drop_prob_lm

    @torch.no_grad()
    def forward(self, xt, fc_feats, att_feats, p_att_feats, state, att_masks=None):
        h, c = state
        N, R = h[0].shape
        E = xt.shape[1]
        ws = ops.default_workspace(xt.device)
        f = lambda t: t.detach().float().contiguous()              # noqa: E731
        xt, fc_feats, att_feats, p_att_feats = f(xt), f(fc_feats), f(att_feats), f(p_att_feats)
        h_lang_prev, h_att_prev, c_att_prev, c_lang_prev = f(h[1]), f(h[0]), f(c[0]), f(c[1])
        a, l, at = self.att_lstm, self.lang_lstm, self.attention
        ld = 2 * R + E
        W = a.weight_ih.detach()
        # att_lstm input = cat([prev_h, fc_feats, xt]) (AttModel.py:626): three K segments of W_ih + W_hh, read in place
        splits = ops.gemm([(h_lang_prev, R, W, ld, R, 1), (fc_feats, R, (W, R), ld, R, 1), (xt, E, (W, 2 * R), ld, E, 1),
                           (h_att_prev, R, a.weight_hh.detach(), R, R, 1)], N, 4 * R, ws.buf, ws=ws, defer_reduce=True)
        h_att, c_att, _, _ = ops.lstm_cell_fwd(ws.slabs, splits, a.bias_ih.detach(), a.bias_hh.detach(), c_att_prev)
        att_h = ops.linear(h_att, at.h2att.weight.detach(), at.h2att.bias.detach(), ws=ws)
        ctx, _ = ops.attention_fwd(att_h, p_att_feats, att_feats, None if att_masks is None else f(att_masks),
                                   at.alpha_net.weight.detach().reshape(-1).contiguous(), at.alpha_net.bias.detach(), 1)
        Wl = l.weight_ih.detach()
        splits = ops.gemm([(ctx, R, Wl, 2 * R, R, 1), (h_att, R, (Wl, R), 2 * R, R, 1),
                           (h_lang_prev, R, l.weight_hh.detach(), R, R, 1)], N, 4 * R, ws.buf, ws=ws, defer_reduce=True)
        h_lang, c_lang, _, _ = ops.lstm_cell_fwd(ws.slabs, splits, l.bias_ih.detach(), l.bias_hh.detach(), c_lang_prev)
        out = nn.functional.dropout(h_lang, self.drop_prob_lm, self.training)
        return out, (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang]))


class AttModel(CaptionModel):
    def __init__(self, opt):
        super().__init__()
        self.vocab_size = opt.vocab_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = getattr(opt, 'max_length', 20) or opt.seq_length     # AttModel.py:60
        self.fc_feat_size = opt.fc_feat_size
        self.att_feat_size = opt.att_feat_size
        self.att_hid_size = opt.att_hid_size
        self.bos_idx = getattr(opt, 'bos_idx', 0)
        self.eos_idx = getattr(opt, 'eos_idx', 0)
        self.pad_idx = getattr(opt, 'pad_idx', 0)
        self.unk_idx = getattr(opt, 'unk_idx', None)
        if (self.bos_idx, self.eos_idx, self.pad_idx) != (0, 0, 0):
            raise NotImplementedError('capmi kernels assume bos=eos=pad=0 (the reference default, AttModel.py:65-67)')
        if getattr(opt, 'use_bn', 0):
            raise NotImplementedError('use_bn is outside the BASELINE configs')
        if getattr(opt, 'logit_layers', 1) != 1:
            raise NotImplementedError('logit_layers > 1 is broken in the reference itself (AttModel.py:92)')
        self.ss_prob = 0.0
        self.embed = nn.Sequential(nn.Embedding(self.vocab_size + 1, self.input_encoding_size), nn.ReLU(),
                                   nn.Dropout(self.drop_prob_lm))
        self.fc_embed = nn.Sequential(nn.Linear(self.fc_feat_size, self.rnn_size), nn.ReLU(), nn.Dropout(self.drop_prob_lm))
        self.att_embed = nn.Sequential(nn.Linear(self.att_feat_size, self.rnn_size), nn.ReLU(),
                                       nn.Dropout(self.drop_prob_lm))
        self.logit = nn.Linear(self.rnn_size, self.vocab_size + 1)
        self.ctx2att = nn.Linear(self.rnn_size, self.att_hid_size)
        self.vocab = opt.vocab
        self._flat = None
        self._rng_calls = 0
        self._last_rollout = None

    # ------------------------------------------------------------------ parameter plumbing
    @property
    def _param_names(self):
        return self._param_name_list()

    def flatten_parameters_(self):
        """Move all parameters into one flat buffer (+ flat grads, Adam state).  Call after .cuda()."""
        from imagecaptioning.pytorch_amd.flat import FlatParams
        self._flat = FlatParams(self)
        return self._flat

    def _grad_targets(self, P):
        if self._flat is not None:
            return self._flat.grad_views
        return {k: torch.empty_like(v) for k, v in P.items()}

    def _device_check(self, t):
        if not t.is_cuda:
            raise CapmiError('the capmi backend runs on a HIP device only (got a %s tensor); there is no CPU path'
                             % t.device.type)

    def init_hidden(self, bsz):
        w = self.logit.weight
        return (w.new_zeros(self.num_layers, bsz, self.rnn_size), w.new_zeros(self.num_layers, bsz, self.rnn_size))

    def _next_seed(self):
        self._rng_calls += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._rng_calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def _dropout_masks(self, B, K, N, T, dev, eval_rows_from=None):
        """Philox keep-masks for one rollout (dropout is ON in train mode, also while sampling:
        loss_wrapper.py:63), all four in ONE launch.  Returns dict of pre-scaled masks or {} in eval mode / p == 0.
        eval_rows_from: caption rows >= this index run in eval mode (mask 1.0): the greedy rows of the fused SCST rollout."""
        p = self.drop_prob_lm
        if not self.training or p <= 0:
            return {}
        seed = self._next_seed()
        R, E = self.rnn_size, self.input_encoding_size
        fc, att, xt, out = ops.dropout_masks([((B, R), 0, None, dev), ((B, K, R), 1 << 40, None, dev),
                                              ((T, N, E), 2 << 40, eval_rows_from, dev),
                                              ((T, N, R), 3 << 40, eval_rows_from, dev)], p, seed)
        return dict(drop_fc=fc, drop_att=att, drop_xt=xt, drop_out=out)

    def _run(self, cfg, fc_feats, att_feats, att_masks):
        self._device_check(fc_feats)
        params = self._param_list()
        fc_feats = fc_feats.float().contiguous()
        att_feats = att_feats.float().contiguous()
        if att_masks is not None:
            att_masks = att_masks.float().contiguous()
        cfg['_sink'] = sink = sparse_logp.LogpSink()
        seq, logp = _RolloutFn.apply(self, cfg, fc_feats, att_feats, att_masks, *params)
        return seq, sparse_logp.attach(logp, sink)

    # ------------------------------------------------------------------ reference API
    def _prepare_feature(self, fc_feats, att_feats, att_masks):
        """AttModel.py:114-124 (eval-mode numerics; training dropout is applied inside the rollouts)."""
        P = {k: v.detach() for k, v in self._named_param_list()}
        pr = engine.prepare(P, fc_feats.float().contiguous(), att_feats.float().contiguous(),
                            None if att_masks is None else att_masks.float())
        return pr.fc, pr.att, pr.p_att, pr.att_masks

    def _forward(self, fc_feats, att_feats, seq, att_masks=None):
        """Teacher-forced log-probs [N,T,V1] (AttModel.py:126-164)."""
        self._device_check(fc_feats)
        B = fc_feats.size(0)
        if seq.ndim == 3:
            seq = seq.reshape(-1, seq.shape[2])
        seq = seq.long().contiguous()
        N, T = seq.shape
        n = N // B
        # AttModel.py:158-159: stop at the first all-pad column.  The labels are an INPUT, so this is one
        # host decision per batch made before anything is enqueued, not a per-step sync.
        colsum = seq[:, 1:].sum(0)
        zero_cols = (colsum == 0).nonzero()
        T_eff = int(zero_cols[0]) + 1 if zero_cols.numel() else T
        K = clip_len(att_masks, att_feats.shape[1])
        cfg = dict(n=n, T=T_eff, L=T, mode='forced', forced=seq, teacher=True)
        cfg.update(self._dropout_masks(B, K, N, T_eff, fc_feats.device))
        if self.training and self.ss_prob > 0.0:
            # scheduled sampling (AttModel.py:145-154): from step 1 on each row feeds, with probability ss_prob, a draw from
            # the model's previous distribution instead of the ground-truth word.  The coin flips are made here for all
            # steps at once (they do not depend on the model), the draws happen inside the rollout.
            coin = self._ss_coin if getattr(self, '_ss_coin', None) is not None else \
                torch.rand(T_eff, N, device=fc_feats.device) < self.ss_prob            # _ss_coin / _ss_gumbel: test hooks
            cfg['ss_mode'] = torch.where(coin, 1, 2).to(torch.uint8).contiguous()
            cfg['seed'] = self._next_seed()
            if getattr(self, '_ss_gumbel', None) is not None:
                cfg['gumbel'] = self._ss_gumbel
        _, logp = self._run(cfg, fc_feats, att_feats, att_masks)
        return logp

    def _sample(self, fc_feats, att_feats, att_masks=None, opt={}):
        """Greedy / sampling rollout (AttModel.py:258-352)."""
        self._device_check(fc_feats)
        sample_method = opt.get('sample_method', 'greedy')
        beam_size = opt.get('beam_size', 1)
        temperature = opt.get('temperature', 1.0)
        sample_n = int(opt.get('sample_n', 1))
        group_size = opt.get('group_size', 1)
        if beam_size > 1 and sample_method in ('greedy', 'beam_search'):
            return self._sample_beam(fc_feats, att_feats, att_masks, opt)
        from imagecaptioning.pytorch_amd import decode
        if decode.wants_options(opt):
            # _diverse_sample (AttModel.py:270-271) / decoding constraints (:293-330): host-stepped, same kernels
            def make(rows):
                from imagecaptioning.pytorch_amd.step import UpDownStepper
                P = {k: v.detach() for k, v in self._named_param_list()}
                pr = engine.prepare(P, fc_feats.float().contiguous(), att_feats.float().contiguous(),
                                    None if att_masks is None else att_masks.float())
                return UpDownStepper(P, pr, rows)
            return self._sample_with_options(make, fc_feats.size(0), opt)
        mode, temperature, top_k, top_p = parse_sample_method(sample_method, temperature)
        B = fc_feats.size(0)
        N = B * sample_n
        L = self.seq_length
        K = clip_len(att_masks, att_feats.shape[1])
        cfg = dict(n=sample_n, T=L, L=L, mode=mode, temperature=temperature, seed=self._next_seed(), top_k=top_k, top_p=top_p)
        if not opt.get('output_logsoftmax', 1):   # AttModel.py:171-175, 265: seqLogprobs holds the LOGITS (margin structure losses)
            cfg['raw_logits'] = True
        cfg.update(self._dropout_masks(B, K, N, L, fc_feats.device))
        forced = opt.get('_forced_seq')           # test hook: teacher-force a sampled sequence
        if forced is not None:
            cfg.update(mode='forced', forced=forced.long().contiguous())
        gumbel = opt.get('_gumbel')               # test hook: injected noise
        if gumbel is not None:
            cfg['gumbel'] = gumbel
        seq, logp = self._run(cfg, fc_feats, att_feats, att_masks)
        return seq, logp

    def scst_rollouts(self, fc_feats, att_feats, att_masks=None, sample_n=5, temperature=1.0, _gumbel=None):
        """Both rollouts of one self-critical step in ONE pass (MI355X-first: the decode GEMMs run on
        64-row MFMA tiles, so the B greedy-baseline rows ride along with the B*n sampled rows for free and
        the weights are streamed from HBM once instead of twice).

        Semantics of loss_wrapper.py:57-68 are kept per row: sampled rows = train mode (dropout p, categorical
        sampling, gradient), greedy rows = eval mode (no dropout, arg-max, no gradient).
        Returns (greedy_res [B,L], gen_result [B*n,L], sample_logprobs [B*n,L,V1])."""
        self._device_check(fc_feats)
        B, n, L = fc_feats.size(0), int(sample_n), self.seq_length
        N, dev = B * n, fc_feats.device
        K = clip_len(att_masks, att_feats.shape[1])
        was_training = self.training
        self.train()                               # dropout masks for the sampled rows
        cfg = dict(n=n, T=L, L=L, mode='sample', temperature=temperature, seed=self._next_seed(), fused_greedy=True)
        cfg.update(self._dropout_masks(B, K, N + B, L, dev, eval_rows_from=N))     # greedy rows: eval mode
        self.train(was_training)
        cfg['row_img'], cfg['row_mode'] = self._fused_maps(B, n, dev)
        if _gumbel is not None:
            cfg['gumbel'] = _gumbel
        seq, logp = self._run(cfg, fc_feats, att_feats, att_masks)
        gen = seq[:N]
        gen._capmi_all = seq             # sampled rows first, greedy rows behind them: the reward kernel scores them in place
        return seq[N:], gen, sparse_logp.attach_rows(logp[:N], logp)

    def _fused_maps(self, B, n, dev):
        """row -> feature image and row -> mode of the fused SCST rollout (cached per shape: built once, not per step)"""
        cache = self.__dict__.setdefault('_fused_map_cache', {})
        key = (B, n, str(dev))
        if key not in cache:
            N = B * n
            row_img = torch.cat([torch.arange(N, device=dev) // n, B + torch.arange(B, device=dev)]).to(torch.int32)
            row_mode = torch.cat([torch.ones(N, dtype=torch.uint8, device=dev), torch.zeros(B, dtype=torch.uint8, device=dev)])
            hyp_img = torch.cat([torch.arange(N, device=dev) // n, torch.arange(B, device=dev)]).to(torch.int32)
            cache[key] = (row_img, row_mode, hyp_img)
        return cache[key][:2]

    def _sample_beam(self, fc_feats, att_feats, att_masks=None, opt={}):
        from imagecaptioning.pytorch_amd.beam import sample_beam
        return sample_beam(self, fc_feats, att_feats, att_masks, opt)

    def get_logprobs_state(self, it, fc_feats, att_feats, p_att_feats, att_masks, state, output_logsoftmax=1):
        """One decoder step on already prepared features (AttModel.py:166-176); used by beam search."""
        from imagecaptioning.pytorch_amd.step import updown_step
        return updown_step(self, it, fc_feats, att_feats, p_att_feats, att_masks, state, output_logsoftmax)


class UpDownModel(AttModel):
    """AttModel.py:875-879."""

    def __init__(self, opt):
        super().__init__(opt)
        self.num_layers = 2
        self.core = UpDownCore(opt)
