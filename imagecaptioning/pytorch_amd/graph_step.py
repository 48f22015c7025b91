"""One optimisation step of the trainer (reference tools/train.py:185-196: LossWrapper forward, backward, value clip, Adam) as ONE
object -- stepped launch by launch from Python, or captured once per input shape into a hipGraph and replayed.

Why: the Transformer and AoA steps are 450-600 launches issued from Python (8-10 ms of host time against 13 ms of device time per
Transformer XE step); in 8-GPU data parallel every rank waits for the slowest rank's interpreter at every all-reduce.  A replayed
graph costs the host one call and runs the device back to back.

What a graph would freeze is read from device memory instead (ops.StepState, capmi.h capmi_step_state):
  * the dropout / sampling random streams: every seed-taking kernel offsets its seed argument by the record's `epoch` word while
    the record is bound (capmi_rng_bind_epoch); the model's per-iteration seed sequence restarts at every step, the epoch moves on;
  * Adam's step count, bias corrections and learning rate (capmi_adam_step_dyn; capmi_step_set_lr when the schedule moves).
`capmi_step_advance` is the first launch of every iteration.  The STEPPED path of this class runs exactly the same launches under
the same record, so stepped and captured training produce the same numbers bit for bit (tests/test_graph_step_gpu.py) -- the
captured path is an issue-side optimisation, not a different computation.

N > 1 ranks: the graph ends with the gradients in the flat buffer; the ONE RCCL all-reduce and the clip+Adam launch follow it on the
same stream, un-captured (north_star: a single all-reduce per step).

Inputs of a replay are copied into the buffers the capture recorded (device-to-device, on the stream, in front of the graph).
A first batch of a new shape runs stepped (that run also warms every lazily built cache the capture would otherwise record but not
execute); the second one of that shape is captured.
"""
import sys

import torch

from . import ops
from .captioning.utils import rewards

TENSOR_KEYS = ('fc_feats', 'att_feats', 'labels', 'masks', 'att_masks')


class _Entry:
    __slots__ = ('graph', 'static', 'loss', 'out', 'seen')

    def __init__(self):
        self.graph, self.static, self.loss, self.out, self.seen = None, None, None, None, 0


class TrainStep:
    def __init__(self, lw, flat, opt, device, world=1, graph=None, max_graphs=6, all_reduce=None, capture_after=1):
        """lw: LossWrapper; flat: the model's FlatParams; graph: None = the model's own default (`graph_step` attribute: Transformer
        and AoA), True / False force it.  all_reduce: callable returning the gradient scale (default flat.all_reduce when world > 1)."""
        import os
        self.lw, self.flat, self.opt, self.dev, self.world = lw, flat, opt, torch.device(device), int(world)
        self.model = lw.model
        env = os.environ.get('CAPMI_GRAPH_STEP')
        if graph is None:
            graph = bool(getattr(self.model, 'graph_step', False)) if env is None else env != '0'
        self.graph = bool(graph)
        self.max_graphs, self.capture_after = max_graphs, capture_after
        self.entries = {}
        self.state = ops.StepState(self.dev, adam_step=flat.step_count, epoch=int(getattr(self.model, '_rng_calls', 0)))
        self.one = torch.ones((), dtype=torch.float32, device=self.dev)
        self.all_reduce = all_reduce if all_reduce is not None else (flat.all_reduce if self.world > 1 else None)
        self.pool = None
        self.replays = self.captures = self.stepped = 0
        self.failed = None
        self._idx = {}

    # ------------------------------------------------------------------ the iteration itself
    def _adam(self, scale):
        o = self.opt
        clip = o.grad_clip_value if getattr(o, 'grad_clip_mode', 'value') == 'value' else 0.0
        ops.adam_step_dyn(self.flat.flat, self.flat.grad, self.flat.exp_avg, self.flat.exp_avg_sq, self.state, o.optim_alpha,
                          o.optim_beta, o.optim_epsilon, o.weight_decay, clip, scale)

    def _body(self, t, sc_flag, struc_flag, with_adam):
        """advance the step record, forward + loss + backward into the flat gradient buffer (+ clip/Adam when no collective follows)"""
        o, flat = self.opt, self.flat
        self.state.advance(o.optim_alpha, o.optim_beta)
        self.model._rng_calls = 0          # the seeds of one iteration are a fixed sequence; the bound epoch word moves the streams
        B = t['att_feats'].shape[0] if t.get('att_feats') is not None else t['fc_feats'].shape[0]
        idx = self._idx.get(B)
        if idx is None:
            idx = self._idx[B] = torch.arange(B)
        out = self.lw(t.get('fc_feats'), t.get('att_feats'), t.get('labels'), t.get('masks'), t.get('att_masks'), t.get('gts'), idx,
                      sc_flag, struc_flag, False)
        loss = out['loss']
        if loss.dim():
            loss = loss.mean()
        flat.zero_grad()
        flat.expect_backwards(2 if (struc_flag and 0 < getattr(o, 'structure_loss_weight', 1) < 1) else 1)
        loss.backward(gradient=self.one)
        flat.collect_grads()
        if with_adam:
            self._adam(1.0)
        return loss.detach(), {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}

    # ------------------------------------------------------------------ inputs
    @staticmethod
    def _signature(data, sc_flag, struc_flag):
        sig = [bool(sc_flag), bool(struc_flag)]
        for k in TENSOR_KEYS:
            t = data.get(k)
            sig.append(None if t is None else (tuple(t.shape), t.dtype, getattr(t, '_capmi_kmax', None) if k == 'att_masks' else None))
        packed = getattr(data.get('gts'), 'packed', None)
        if packed is not None:
            sig.append(tuple(None if x is None else tuple(x.shape) for x in (packed[0], packed[1], getattr(packed, 'cooked', None))))
        else:
            sig.append(None if data.get('gts') is None else len(data['gts']))
        return tuple(sig)

    def _make_static(self, data, need_gts):
        st = {}
        for k in TENSOR_KEYS:
            t = data.get(k)
            st[k] = None if t is None else t.clone()
            if t is not None and hasattr(t, '_capmi_kmax'):
                st[k]._capmi_kmax = t._capmi_kmax
        gts = data.get('gts')
        if need_gts and gts is not None:
            packed = rewards.pack_gts(gts).packed
            cooked = getattr(packed, 'cooked', None)
            from .ciderd import PackedRefs
            g = rewards.GtsBatch([None] * len(gts))
            g.packed = PackedRefs(packed[0].clone(), packed[1].clone(), None if cooked is None else cooked.clone())
            st['gts'] = g
        else:
            st['gts'] = gts
        return st

    @staticmethod
    def _fill_static(st, data, need_gts):
        for k in TENSOR_KEYS:
            if st[k] is not None:
                st[k].copy_(data[k], non_blocking=True)
        if need_gts and data.get('gts') is not None:
            packed = rewards.pack_gts(data['gts']).packed
            dst = st['gts'].packed
            dst[0].copy_(packed[0], non_blocking=True)
            dst[1].copy_(packed[1], non_blocking=True)
            if dst.cooked is not None:
                dst.cooked.copy_(packed.cooked, non_blocking=True)

    # ------------------------------------------------------------------ public
    def __call__(self, data, sc_flag=False, struc_flag=False, lr=None, drop_worst_flag=False, force_stepped=False):
        """-> (loss 0-dim device tensor, dict of the LossWrapper's tensor outputs).  `data`: the loader's batch dict on the device.
        force_stepped: run THIS iteration launch by launch even when a graph of its shape exists (same numbers; the launches of a
        replayed graph cannot carry the per-launch HIP events bench.py's roofline sample needs)."""
        o = self.opt
        self.state.set_lr(o.learning_rate if lr is None else lr)
        self.flat.step_count += 1
        multi = self.all_reduce is not None
        graphable = self.graph and not force_stepped and self.failed is None and not drop_worst_flag and getattr(self.model, 'ss_prob', 0.0) == 0.0 \
            and self.flat.on_grads_ready is None
        ent = None
        if graphable:
            sig = self._signature(data, sc_flag, struc_flag)
            ent = self.entries.get(sig)
            if ent is None:
                if len(self.entries) >= self.max_graphs:
                    self.entries.pop(next(iter(self.entries)))
                ent = self.entries[sig] = _Entry()
        need_gts = bool(sc_flag or struc_flag)
        if ent is not None and ent.graph is None and self.captures >= 8 and self.replays < 4 * self.captures:
            # shape churn (ragged region counts: a new input shape almost every batch): captures cost more than their replays save
            self.failed = 'input shapes change too often (%d captures for %d replays): stepping launch by launch' % (self.captures, self.replays)
            print('capmi: ' + self.failed, file=sys.stderr, flush=True)
            ent = None
        if ent is not None and ent.graph is None and ent.seen >= self.capture_after:
            self._capture(ent, data, sc_flag, struc_flag, need_gts, not multi)
        if ent is not None and ent.graph is not None:
            self._fill_static(ent.static, data, need_gts)
            ent.graph.replay()
            self.replays += 1
            loss, out = ent.loss, ent.out
        else:
            if ent is not None:
                ent.seen += 1
            if drop_worst_flag:
                return self._stepped_drop_worst(data, sc_flag, struc_flag)
            # (capture_scratch: zero-padded scratch the step asks for is filed where a later capture of this step looks for it)
            with self.state.bound(), ops.capture_scratch():
                loss, out = self._body(data, sc_flag, struc_flag, with_adam=not multi)
            self.stepped += 1
        if multi:
            self._adam(self.all_reduce())
        return loss, out

    def _stepped_drop_worst(self, data, sc_flag, struc_flag):
        """tools/train.py:187-191 (drop_worst_after): the mean over the rows with the lowest loss -- a data-dependent row count only
        through opt.drop_worst_rate, kept on the stepped path"""
        o, flat = self.opt, self.flat
        with self.state.bound():
            self.state.advance(o.optim_alpha, o.optim_beta)
            self.model._rng_calls = 0
            out = self.lw(data.get('fc_feats'), data.get('att_feats'), data.get('labels'), data.get('masks'), data.get('att_masks'),
                          data.get('gts'), torch.arange(len(data['gts'])) if data.get('gts') is not None else None, sc_flag, struc_flag,
                          True)
            rows = out['loss']
            loss = torch.topk(rows, k=int(rows.shape[0] * (1 - o.drop_worst_rate)), largest=False)[0].mean()
            flat.zero_grad()
            loss.backward()
            flat.collect_grads()
        self._adam(self.all_reduce() if self.all_reduce is not None else 1.0)
        self.stepped += 1
        return loss.detach(), {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}

    def _capture(self, ent, data, sc_flag, struc_flag, need_gts, with_adam):
        ops.reserve_pinned_arena()
        static = self._make_static(data, need_gts)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with self.state.bound():
                # (thread_local: the prefetcher's copy thread and RCCL's watchdog thread keep making runtime calls while this thread
                #  captures -- event queries, pinned allocations -- which the default global mode would turn into capture errors)
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode='thread_local'):
                    loss, out = self._body(static, sc_flag, struc_flag, with_adam)
        except Exception as e:          # a family whose step synchronises with the host cannot be captured: keep stepping, say so once
            self.failed = '%s: %s' % (type(e).__name__, e)
            print('capmi: the training step could not be captured into a hipGraph (%s); stepping launch by launch'
                  % self.failed.split('\n')[0][:200], file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            return
        if self.pool is None:
            self.pool = g.pool()
        ent.graph, ent.static, ent.loss, ent.out = g, static, loss, out
        self.captures += 1

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        return {'epoch': int(self.state.read().epoch), 'adam_step': int(self.flat.step_count)}
