"""Flat parameter / gradient storage.

All parameters of a model live in ONE contiguous fp32 device buffer and all gradients in another:
  * the optimizer step is one fused clip+Adam launch over the flat buffer (capmi_adam_step);
  * data-parallel training moves the gradient with a SINGLE RCCL all-reduce per step,
    instead of nn.DataParallel's per-step parameter broadcast + gradient reduce (train.py:86-88);
  * the BPTT kernels write gradients straight into views of the flat buffer.
``state_dict()`` is unchanged: every nn.Parameter keeps its name and shape, its storage is a view.
"""
import torch

from . import ops


class FlatParams:
    def __init__(self, module):
        params = [(n, p) for n, p in module.named_parameters()]
        if not params:
            raise ValueError('module has no parameters')
        dev = params[0][1].device
        self.names = [n for n, _ in params]
        self.params = [p for _, p in params]
        sizes = [p.numel() for p in self.params]
        # 16-byte align every segment so the MFMA loaders can use 16-byte vector loads.  Segments are laid out in the order of
        # model.parameters() except for the module's `_flat_groups()`: lists of parameter names that must sit back to back so
        # that ONE GEMM can take them as a single operand (e.g. [Wq; Wk; Wv] of an attention block as a [3D, D] matrix, r4).
        # names / params / offsets stay indexed in model.parameters() order (optimizer.pth of the reference relies on it).
        index = {n: i for i, n in enumerate(self.names)}
        order, placed = [], set()
        groups = module._flat_groups() if hasattr(module, '_flat_groups') else []
        first_of = {}
        for g in groups:
            g = [n for n in g if n in index]
            if len(g) > 1 and not placed.intersection(g):
                first_of[min(index[n] for n in g)] = g
                placed.update(g)
        for i, n in enumerate(self.names):
            if i in first_of:
                order += [index[m] for m in first_of[i]]
            elif n not in placed:
                order.append(i)
        assert sorted(order) == list(range(len(self.names)))
        self.offsets, off = [0] * len(sizes), 0
        for i in order:
            self.offsets[i] = off
            off += (sizes[i] + 3) // 4 * 4
        self.used = off
        self.total = (off + 63) // 64 * 64          # equal 16-byte-aligned shards for up to 16 ranks (sharded_step)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad_views = {}
        for n, p, o in zip(self.names, self.params, self.offsets):
            view = self.flat[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            self.grad_views[n] = self.grad[o:o + p.numel()].view_as(p)
        self._view_ptrs = [self.grad_views[n].data_ptr() for n in self.names]       # (collect_grads)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self.on_grads_ready = None      # set by begin_overlap(); native backwards call it per finished bucket
        self._bw_expected, self._bw_seen = 1, 0

    # ---- optimizer state (the reference's optimizer.pth, misc.py:87-102 / tools/train.py:112-119) --------------------
    def _natural_offsets(self):
        """segment offsets of the layout WITHOUT groups (model.parameters() order): what checkpoints written before round 4 used"""
        out, off = {}, 0
        for n, p in zip(self.names, self.params):
            out[n] = off
            off += (p.numel() + 3) // 4 * 4
        return out

    def state_dict(self):
        return {'exp_avg': self.exp_avg.detach().cpu().clone(), 'exp_avg_sq': self.exp_avg_sq.detach().cpu().clone(),
                'step_count': int(self.step_count), 'total': int(self.total),
                'offsets': {n: int(o) for n, o in zip(self.names, self.offsets)}}      # r4: the layout the moments were saved in

    def load_state_dict(self, sd):
        if int(sd['total']) != self.total:
            raise ValueError('optimizer state is for %d flat elements, the model has %d' % (int(sd['total']), self.total))
        mine = {n: int(o) for n, o in zip(self.names, self.offsets)}
        theirs = sd.get('offsets') or self._natural_offsets()       # (no table: a checkpoint of rounds 1-3, natural order)
        if theirs == mine:
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        else:
            # the buffer layout changed (r4 lays out a model's _flat_groups() back to back): move every parameter's moments by NAME
            if set(theirs) != set(mine):
                raise ValueError('optimizer state names do not match this model: %s' % sorted(set(theirs) ^ set(mine))[:4])
            ea, es = sd['exp_avg'], sd['exp_avg_sq']
            for n, p in zip(self.names, self.params):
                k, a, b = p.numel(), theirs[n], mine[n]
                self.exp_avg[b:b + k].copy_(ea[a:a + k])
                self.exp_avg_sq[b:b + k].copy_(es[a:a + k])
        self.step_count = int(sd['step_count'])

    def load_torch_adam_state(self, sd):
        """The reference's optimizer.pth is ``torch.optim.Adam(model.parameters()).state_dict()`` (misc.py:87-102, :112-130):
        {'state': {index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{'params': [index, ...], ...}]}, indices in the
        order of ``model.parameters()`` -- the order of this buffer's segments.  Returns False (nothing loaded) when the state
        does not fit this model."""
        try:
            order = [i for g in sd['param_groups'] for i in g['params']]
            state = sd['state']
        except (KeyError, TypeError):
            return False
        if len(order) != len(self.params):
            return False
        steps = []
        for i, p in zip(order, self.params):
            st = state.get(i)
            if st is None:                         # a parameter that never received a gradient
                continue
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                return False
        for i, p, o in zip(order, self.params, self.offsets):
            st = state.get(i)
            if st is None:
                continue
            self.exp_avg[o:o + p.numel()].view_as(p).copy_(st['exp_avg'])
            self.exp_avg_sq[o:o + p.numel()].view_as(p).copy_(st['exp_avg_sq'])
            steps.append(int(st['step']))
        self.step_count = max(steps) if steps else 0
        return True

    def expect_backwards(self, k):
        """How many native backward passes this optimisation step will run (2 when LossWrapper mixes an XE and a structure
        loss, loss_wrapper.py:25-48).  Gradient buckets may only be handed to the collective by the LAST of them: an earlier
        pass would put buffers in flight that the next pass clones, overwrites and accumulates into."""
        self._bw_expected = max(1, int(k))

    def overlap_allowed(self, stash):
        """True when this backward pass (begin_backward() already called) may announce finished buckets."""
        if self.on_grads_ready is None:
            return False
        if self._bw_seen > self._bw_expected:
            raise RuntimeError('a %d. native backward ran in one step while the bucketed all-reduce is on and only %d were '
                               'declared: gradients already in flight would be overwritten -- call '
                               'flat.expect_backwards(k) before the step (or train without CAPMI_DDP_OVERLAP)'
                               % (self._bw_seen, self._bw_expected))
        # accumulating passes add their stash AFTER the phases ran (end_backward): nothing is final before that
        return self._bw_seen == self._bw_expected and stash is None

    def begin_backward(self):
        """Called by a native backward BEFORE it overwrites the flat gradient views.  Returns a stash of gradients
        that already exist (a second backward pass within one step accumulates, like autograd would)."""
        stash = None
        self._bw_seen += 1
        for n, p in zip(self.names, self.params):
            if p.grad is not None:
                stash = stash or {}
                stash[n] = p.grad.clone()
        return stash

    def end_backward(self, stash=None):
        """Adopt the freshly written flat views as the parameters' .grad -- no copy.  (Handing the views back to
        autograd made AccumulateGrad clone every one of them: ~50 device copies / 0.35 ms per SCST step.)"""
        for n, p in zip(self.names, self.params):
            v = self.grad_views[n]
            if stash is not None and n in stash:
                v.add_(stash[n])
            p.grad = v

    def collect_grads(self):
        """Make the flat gradient buffer authoritative: parameters whose .grad is not already the flat
        view (e.g. produced by torch autograd ops) are copied in; missing grads become zero."""
        for n, p, vp in zip(self.names, self.params, self._view_ptrs):
            g = p.grad
            if g is not None and g.data_ptr() == vp:
                continue                      # autograd adopted the flat view the backward returned: nothing to do
            # (r4: the unconditional `p.grad = v` cost 2.5 us per parameter -- 0.7 ms of idle GPU between the Transformer's last
            #  backward kernel and its Adam launch)
            v = self.grad_views[n]
            if g is None:
                v.zero_()
            else:
                v.copy_(g)
            p.grad = v

    # ---- bucketed all-reduce overlapped with the backward ------------------------------------------------------
    # The BPTT finishes its gradients in a known order (logit layer first, recurrent weights after the time loop).
    # Each finished bucket is a contiguous slice of the flat buffer; its all-reduce is launched asynchronously (RCCL
    # runs it on its own stream, after the launches enqueued so far) while the remaining backward phases compute.
    def begin_overlap(self, group=None, world_size=None):
        import torch.distributed as dist
        self._ov = dict(group=group, ws=world_size or dist.get_world_size(group), works=[], done=[])
        self.on_grads_ready = self._bucket_ready

    def _bucket_ready(self, names):
        """all-reduce the flat ranges covering `names` (all final): parameters adjacent in the buffer share a launch."""
        import torch.distributed as dist
        ov = self._ov
        segs = sorted((self.offsets[i], self.offsets[i] + (self.params[i].numel() + 3) // 4 * 4)
                      for i in (self.names.index(n) for n in names))
        lo, hi = segs[0]
        for a, b in segs[1:] + [(None, None)]:
            if a is not None and a == hi:         # adjacent in the BUFFER (layout order, not model.parameters() order)
                hi = b
                continue
            ov['works'].append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=ov['group'], async_op=True))
            ov['done'].append((lo, hi))
            lo, hi = a, b

    def _launch_leftovers(self):
        import torch.distributed as dist
        ov = self._ov
        pos = 0
        for lo, hi in sorted(ov['done']) + [(self.used, self.used)]:      # the padding behind `used` carries no gradient
            if lo > pos:
                ov['works'].append(dist.all_reduce(self.grad[pos:lo], op=dist.ReduceOp.SUM, group=ov['group'], async_op=True))
                ov['done'].append((pos, lo))
            pos = max(pos, hi)

    def finish_overlap(self):
        """Reduce whatever no bucket covered, wait for all collectives (stream-side), return the 1/world scale."""
        ov = self._ov
        self._launch_leftovers()
        for w in ov['works']:
            w.wait()
        self.last_collectives = len(ov['works'])
        ov['works'], ov['done'] = [], []
        return 1.0 / ov['ws']

    def finish_overlap_and_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0):
        """finish_overlap() + adam_step() pipelined per bucket: the fused clip+Adam of a bucket runs as soon as ITS
        all-reduce has landed, while the collectives of the later buckets are still in flight (the update is elementwise,
        so applying it range by range is the same update)."""
        ov = self._ov
        self._launch_leftovers()
        self.step_count += 1
        scale = 1.0 / ov['ws']
        ranges = ov['done'][:len(ov['works'])]
        for w, (lo, hi) in zip(ov['works'], ranges):
            w.wait()                                   # stream-side: the compute stream waits for this bucket only
            ops.adam_step(self.flat[lo:hi], self.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], lr, betas[0],
                          betas[1], eps, weight_decay, clip_value, scale, self.step_count)
        self.last_collectives = len(ov['works'])
        ov['works'], ov['done'] = [], []
        return scale

    def all_reduce(self, group=None, world_size=None):
        """ONE collective for the whole model (RCCL over xGMI when backend == 'nccl'); averages."""
        import torch.distributed as dist
        ws = world_size or dist.get_world_size(group)
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / ws

    def sharded_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0, group=None, adam=None):
        """Reduce-scatter + all-gather instead of all-reduce (the two halves of the collective, with the
        optimizer between them): every rank receives the SUM of ONE 1/G shard of the flat gradient, runs the fused clip+Adam
        on that shard only (1/G of the 28 B/parameter the optimizer streams), and the updated parameter shards are all-gathered.
        Same wire bytes as a ring all-reduce, but the 0.27 ms Adam pass shrinks by G and no rank ever holds -- or clips --
        anything but averaged gradients.  Elementwise update => identical to all_reduce() + adam_step().
        `adam`: test hook replacing ops.adam_step (CPU tests have no HIP device)."""
        import torch.distributed as dist
        ws, rank = dist.get_world_size(group), dist.get_rank(group)
        if self.total % (4 * ws):
            raise ValueError('flat buffer of %d floats does not split into %d aligned shards' % (self.total, ws))
        shard = self.total // ws
        lo, hi = rank * shard, (rank + 1) * shard
        try:
            dist.reduce_scatter_tensor(self.grad[lo:hi], self.grad, op=dist.ReduceOp.SUM, group=group)
        except (RuntimeError, NotImplementedError):        # gloo has no reduce-scatter: same result through an all-reduce
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        self.step_count += 1
        (adam or ops.adam_step)(self.flat[lo:hi], self.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], lr, betas[0],
                                betas[1], eps, weight_decay, clip_value, 1.0 / ws, self.step_count)
        try:
            dist.all_gather_into_tensor(self.flat, self.flat[lo:hi], group=group)
        except (RuntimeError, NotImplementedError):
            dist.all_gather([self.flat[r * shard:(r + 1) * shard] for r in range(ws)], self.flat[lo:hi].clone(), group=group)
        return 1.0 / ws

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0, grad_scale=1.0):
        """clip_grad_value_ (train.py:194-195) + Adam (misc.py:125-126) fused, one launch."""
        self.step_count += 1
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, lr, betas[0], betas[1], eps, weight_decay,
                      clip_value, grad_scale, self.step_count)

    def zero_grad(self):
        self._bw_seen = 0
        for p in self.params:
            p.grad = None
