"""Tensor-level wrappers over the C ABI (include/capmi.h).  Pure plumbing: they check devices/dtypes,
pass ``data_ptr()``s and the current HIP stream, and raise on any non-zero return.  No compute happens
in Python and nothing here falls back to torch ops.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import lib, ptr, check, stream_ptr

_f32 = torch.float32


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.CapmiError('capmi ops need device tensors (got %s)' % t.device)
        if not t.is_contiguous():
            raise _lib.CapmiError('capmi ops need contiguous tensors')


class Workspace:
    """Split-K scratch shared by the GEMM launches of one stream."""

    COUNTER_FLOATS = 16384      # CAPMI_WS_COUNTER_FLOATS: tile tickets live in front of the slabs, zero-initialised

    def __init__(self, device, floats=16 * 1024 * 1024):
        self.buf = torch.zeros(floats + self.COUNTER_FLOATS, dtype=_f32, device=device)

    @property
    def slabs(self):
        return self.buf[self.COUNTER_FLOATS:]

    @property
    def capacity(self):
        return self.buf.numel()


_default_ws = {}


def default_workspace(device):
    key = str(device)
    if key not in _default_ws:
        _default_ws[key] = Workspace(device)
    return _default_ws[key]


def gemm(segs, M, N, out, ldc=None, a_layout=0, b_layout=0, bias=None, bias2=None, row_bias=None, row_bias_div=1,
         mul_mask=None, relu=False, accumulate=False, ws=None, splits=0, defer_reduce=False, a_planes=None, addend=None, stream=None,
         allow_wide=False):
    """segs: list of (A, lda, B, ldb, K, a_row_div) with tensors (or (tensor, element_offset) pairs).
    a_planes: optional list (one uint8 tensor per segment, see planes_from_f32) -- the activations also delivered pre-split,
    staged by LDS-DMA in the M <= 64 decode kernel.
    addend: out = addend + epilogue(...) (a residual stream added without copying it into `out` first; row pitch ldc).
    Returns splits_used."""
    d = _lib.GemmDesc()
    d.nseg = len(segs)
    if a_planes is not None:
        assert len(a_planes) == len(segs)
        for i, t in enumerate(a_planes):
            d.a_planes[i] = t.data_ptr()
    for i, (A, lda, B, ldb, K, div) in enumerate(segs):
        d.seg[i].A = None if A is None else _addr(A)
        d.seg[i].B = None if B is None else _addr(B)
        d.seg[i].lda, d.seg[i].ldb, d.seg[i].K, d.seg[i].a_row_div = lda, ldb, K, div
    d.a_layout, d.b_layout, d.M, d.N = a_layout, b_layout, M, N
    d.C = _addr(out)
    d.ldc = N if ldc is None else ldc
    d.bias, d.bias2, d.row_bias = ptr(bias), ptr(bias2), ptr(row_bias)
    d.row_bias_div = row_bias_div
    d.mul_mask = ptr(mul_mask)
    d.relu, d.accumulate = int(relu), int(accumulate or addend is not None)
    d.addend = ptr(addend)
    if ws is None:
        ws = default_workspace(_dev(out))
    d.partial, d.partial_capacity = ws.buf.data_ptr(), ws.capacity
    d.splits, d.defer_reduce = splits, int(defer_reduce)
    d.allow_wide_deferred = int(allow_wide)      # (deferred GEMMs: nothing runs beside this one on another stream)
    check(lib.capmi_gemm_f32(C.byref(d), stream_ptr() if stream is None else stream), 'capmi_gemm_f32')
    return d.splits_used


_zero_planes = {}
_planes_cache = {}


def zero_planes(device, chunks=1):
    """>= `chunks` all-zero chunk images (the planes of an all-zero activation, e.g. the initial LSTM state), allocated once per device."""
    key = str(device)
    t = _zero_planes.get(key)
    if t is None or t.numel() < chunks * 12288:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('zero planes requested for the first time inside a graph capture (warm the function up first)')
        t = _zero_planes[key] = torch.zeros(max(chunks, 4) * 12288, dtype=torch.uint8, device=device)
    return t


_capture_scratch = [0]


class capture_scratch:
    """`with ops.capture_scratch():` around the eager WARM-UP of a function that is about to be captured into a hipGraph
    (graphs.GraphedDecode): scratch requested inside is filed under the key the capture will ask for, so it is allocated and
    zero-filled eagerly -- a torch.zeros recorded INTO a graph would only run when that graph replays, while the buffer is
    cached process-wide."""

    def __enter__(self):
        _capture_scratch[0] += 1

    def __exit__(self, *a):
        _capture_scratch[0] -= 1
        return False


def planes_scratch(device, tag, nbytes):
    """Zero-filled-once scratch for A planes, cached per (device, tag, size, stream): rows >= M / columns >= K of a planes buffer
    are never written, so a buffer keeps its zero padding across the rollouts that reuse it; concurrent streams get their own.
    Graph captures share one slot that their warm-up fills (capture_scratch); allocating while capturing is refused."""
    capturing = torch.cuda.is_current_stream_capturing()
    slot = 'capture' if (capturing or _capture_scratch[0]) else stream_ptr()
    key = (str(device), tag, int(nbytes), slot)
    t = _planes_cache.get(key)
    if t is None:
        if capturing:
            raise RuntimeError('planes scratch %r requested for the first time inside a graph capture: run the function once '
                               'under ops.capture_scratch() before capturing it' % (tag,))
        t = _planes_cache[key] = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
    return t


def planes_from_f32(x, out=None):
    """A planes (capmi.h capmi_planes_from_f32) of a [M <= 64, K] fp32 matrix (row stride = x.stride(0))."""
    M, K = x.shape
    assert x.is_cuda and x.dtype == _f32 and x.stride(1) == 1 and M <= 64
    if out is None:
        out = torch.zeros(int(lib.capmi_planes_bytes(K)), dtype=torch.uint8, device=x.device)
    check(lib.capmi_planes_from_f32(x.data_ptr(), x.stride(0), M, K, out.data_ptr(), stream_ptr()), 'capmi_planes_from_f32')
    return out


def _addr(x):
    if isinstance(x, tuple):
        t, off = x
        return t.data_ptr() + 4 * off
    return x.data_ptr()


def _dev(x):
    return (x[0] if isinstance(x, tuple) else x).device


class _WsView:
    """A region of a SlabArena with the Workspace interface of gemm()."""

    def __init__(self, buf):
        self.buf = buf

    @property
    def capacity(self):
        return self.buf.numel()


class SlabArena:
    """Bump allocator of split-K workspaces whose reduction is deferred: one region ([tickets | slabs]) per GEMM, the
    same offsets step after step.  A deferred GEMM never self-reduces (gemm_f32.hip: self_reduce requires !defer_reduce), so
    the ticket words in front of each region are layout only -- nothing reads them and nothing has to keep them zero."""

    CHUNK = 128 * 1024 * 1024          # floats (512 MB)

    def __init__(self, device):
        self.dev, self.chunks = device, []
        self.cur = self.off = 0

    def reset(self):
        self.cur = self.off = 0

    def take(self, floats):
        assert floats <= self.CHUNK
        if self.off + floats > self.CHUNK:
            self.cur, self.off = self.cur + 1, 0
        if self.cur == len(self.chunks):
            self.chunks.append(torch.zeros(self.CHUNK, dtype=_f32, device=self.dev))
        return self.chunks[self.cur][self.off:self.off + floats]

    def commit(self, floats):
        self.off += (floats + 63) // 64 * 64


_group_cache = {}


def gemm_group_tn(items, ws=None, cache_key=None):
    """capmi_gemm_group_tn: items = [(dy [K,M], x [K,N], out, accumulate[, ldc, out_off[, colsum]])] -- out_i (+)= dy_i^T x_i for all i in
    ONE persistent launch (+ one small reduction launch for the K-sliced tail).  out: [M,N] contiguous, or -- with ldc / out_off -- the
    tensor whose elements out_off + m * ldc + n receive the product (a column block of a wider gradient).  colsum: optional [M]
    tensor that receives the column sums of dy (the bias gradient), taken by the GEMM's staging waves.
    Returns the per-item splits_used.
    cache_key: any hashable; the ctypes table of an identical item list (same pointers, same shapes) is built once."""
    dev = items[0][2].device
    key = tuple((it[0].data_ptr(), it[1].data_ptr(), it[2].data_ptr(), it[0].shape[0], it[0].shape[1], it[1].shape[1], bool(it[3])) + tuple(it[4:6])
                + ((it[6].data_ptr(),) if len(it) > 6 and it[6] is not None else ()) for it in items)
    hit = _group_cache.get(cache_key) if cache_key is not None else None
    if hit is not None and hit[0] == key:
        arr = hit[1]
    else:
        arr = (_lib.GroupGemm * len(items))()
        for g, it in zip(arr, items):
            dy, x, out, acc = it[:4]
            _chk(dy, x, out)
            K, M = dy.shape
            N = x.shape[1]
            ldc = it[4] if len(it) > 4 and it[4] is not None else N
            off = it[5] if len(it) > 5 else 0
            cs = it[6] if len(it) > 6 else None
            if x.shape[0] != K or ldc < N or off + (M - 1) * ldc + N > out.numel() or (cs is not None and (cs.numel() != M or not cs.is_contiguous())):
                raise _lib.CapmiError('gemm_group_tn: shapes %s^T %s -> %s (ldc %d, offset %d)' % (tuple(dy.shape), tuple(x.shape), tuple(out.shape), ldc, off))
            g.A, g.B, g.C, g.lda, g.ldb, g.ldc, g.K, g.M, g.N, g.accumulate = dy.data_ptr(), x.data_ptr(), out.data_ptr() + 4 * off, M, N, ldc, K, M, N, int(bool(acc))
            g.colsum = None if cs is None else cs.data_ptr()
        if cache_key is not None:
            _group_cache[cache_key] = (key, arr)
    if ws is None:
        ws = default_workspace(dev)
    slabs = ws.buf[Workspace.COUNTER_FLOATS:] if isinstance(ws, Workspace) else ws.buf
    check(lib.capmi_gemm_group_tn(arr, len(items), slabs.data_ptr(), slabs.numel(), stream_ptr()), 'capmi_gemm_group_tn')
    return [g.splits_used for g in arr]


_pinned_arena = {'buf': None, 'off': 0}


def reserve_pinned_arena(nbytes=8 << 20):
    """Pinned host memory for the small tables a graph CAPTURE uploads (pinning inside a capture is refused by the runtime): call
    before capturing.  A slot is handed out once and never recycled -- the captured memcpy node re-reads it at every replay."""
    a = _pinned_arena
    if a['buf'] is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('the pinned table arena must be reserved before a graph capture starts')
        a['buf'], a['off'] = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True), 0
    return a


def upload_bytes(raw, device):
    """host bytes -> uint8 device tensor on the current stream, valid under graph capture"""
    import numpy as np
    n = len(raw)
    if not torch.cuda.is_current_stream_capturing():
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).pin_memory().to(device, non_blocking=True)
    a = _pinned_arena
    if a['buf'] is None or a['off'] + n > a['buf'].numel():
        raise RuntimeError('pinned table arena missing or full (%d bytes wanted): ops.reserve_pinned_arena() before the capture' % n)
    slot = a['buf'][a['off']:a['off'] + n]
    a['off'] += (n + 63) // 64 * 64
    slot.numpy()[:] = np.frombuffer(raw, dtype=np.uint8)
    t = torch.empty(n, dtype=torch.uint8, device=device)
    check(lib.capmi_upload_async(t.data_ptr(), slot.data_ptr(), n, stream_ptr()), 'capmi_upload_async')
    return t


class DeferredGrads:
    """Parameter gradients of a layer-by-layer backward that nothing reads before the optimizer: the weight-gradient GEMMs
    leave their K-slice slabs in a SlabArena and the bias-gradient column sums are only recorded; batched launches finish them
    (Transformer XE bs64: 98 + 162 launches and 160 zero-fills less per step).
    r4: all of it runs on a SIDE STREAM beside the dX / attention / LayerNorm chain of the backward -- nothing reads these results
    before flush(), and every kernel of the chain leaves CUs idle at its tail (212 or 636 tiles of a d_model-wide GEMM on 256 CUs)
    or altogether (MHA, LayerNorm, masks).  The GEMMs follow their operands by an event; the reductions and column sums go out in
    batches of SIDE_BATCH items behind them instead of as two launches at the very end of the backward.  Transformer XE 15.3 ->
    14.3 ms per step with the GEMMs alone (A/B inside one gpurun call).
    r6: superseded as the default by the grouped launch (see __init__); CAPMI_DW_STREAM=1 opts back in."""

    _arenas = {}
    uploads = 0
    SIDE_BATCH = 64            # (batches of 12 vs one flush at the end: 14.37 vs 14.44 ms per Transformer XE step -- and 46 more launches; 64: +6)

    def __init__(self, device):
        self.dev = device
        key = str(device)
        if key not in self._arenas:
            self._arenas[key] = {'arena': SlabArena(device), 'tables': {}}
        self.state = self._arenas[key]
        self.arena = self.state['arena']
        self.arena.reset()
        self.red, self.col, self.keep = [], [], []
        self.red_side, self.col_side, self.side_batches, self.synced = [], [], 0, None
        self.side = None
        # r5: the side stream was worth 2-5 % of the Transformer step beside one launch + deferred reduction per GEMM (profiles/
        # r05_fat_gemm_wide.md section 7).  r6: the GROUPED launch at flush() beats it in both issue modes (stepped 20.81 vs 20.98 ms,
        # captured 20.70 vs 22.78 / 23.24 with the side stream forked into the capture; gpurun_out/r6d, r6e) -- the side stream is
        # opt-in now (CAPMI_DW_STREAM=1; inside a graph capture additionally CAPMI_DW_STREAM_CAPTURE=1), and the stepped and the
        # captured step issue exactly the same launches
        if os.environ.get('CAPMI_DW_STREAM', '0') == '1' and torch.cuda.is_available() and \
                (not torch.cuda.is_current_stream_capturing() or os.environ.get('CAPMI_DW_STREAM_CAPTURE', '0') == '1'):
            if 'side' not in self.state:
                self.state['side'] = torch.cuda.Stream(device=device)
                self.state['events'] = []
            self.side, self.ev_pool, self.ev_used = self.state['side'], self.state['events'], 0
        # nothing runs beside the deferred GEMMs when there is no side stream: the planner may give them 256 x 128 tiles too -- said
        # per call (capmi_gemm_desc.allow_wide_deferred; r5 flipped the process-wide capmi_gemm_set_policy flag here, which two live
        # instances could leave stuck: ADVICE r5)
        # r6: without a side stream the weight-gradient GEMMs are only RECORDED and go out at flush() as one grouped persistent launch
        # (capmi_gemm_group_tn); CAPMI_DW_GROUP=0 keeps one launch + deferred reduction per GEMM
        self.group = [] if (self.side is None and os.environ.get('CAPMI_DW_GROUP', '1') != '0') else None

    def _side_follows_main(self):
        """the side stream waits for everything enqueued on the current stream so far"""
        if self.ev_used == len(self.ev_pool):
            self.ev_pool.append(torch.cuda.Event())
        ev = self.ev_pool[self.ev_used]
        self.ev_used += 1
        ev.record()
        self.side.wait_event(ev)

    def dw(self, dy, x, out, final=True, ldc=None, out_off=0, accumulate=False, colsum_out=None):
        """out[M,N] (+)= dy[K,M]^T x[K,N], reduction deferred.  out: contiguous [M,N], or with ldc / out_off a column block of a wider
        gradient (elements out_off + m * ldc + n).  final: nobody writes `dy` after this call (a running gradient accumulator that the
        caller keeps adding to must be read in stream order: no side stream, no grouping).  accumulate: the caller has ALREADY
        written the addend into `out` (in stream order).  colsum_out: the bias gradient (column sums of dy) that goes with this weight
        gradient -- in a grouped launch the GEMM's staging waves take it from the operand they stage (no second pass over dy)."""
        _chk(dy, x, out)
        K, M = dy.shape
        N = x.shape[1]
        ldc = N if ldc is None else ldc
        cf = Workspace.COUNTER_FLOATS
        if self.group is not None and final:
            fold = colsum_out is not None and os.environ.get('CAPMI_GROUP_COLSUM', '1') != '0'
            self.group.append((dy, x, out, accumulate, ldc, out_off, colsum_out if fold else None))
            if colsum_out is not None and not fold:
                self.colsum(dy, colsum_out)
            return
        if colsum_out is not None:
            if final:
                self.colsum(dy, colsum_out)
            else:
                colsum(dy, out=colsum_out)
        if cf + 2 * M * N > SlabArena.CHUNK:            # a gradient this large needs no K split to fill the chip
            gemm([(dy, M, x, N, K, 1)], M, N, (out, out_off), ldc=ldc, a_layout=1, b_layout=1, accumulate=accumulate)
            return
        region = self.arena.take(min(cf + 16 * M * N, SlabArena.CHUNK))      # (the GEMM limits its K split to the region)
        on_side = self.side is not None and final
        if on_side:
            self._side_follows_main()         # dy and x are complete on the main stream here
            self.synced = dy
            self.keep.extend((dy, x))
        splits = gemm([(dy, M, x, N, K, 1)], M, N, out, a_layout=1, b_layout=1, ws=_WsView(region), defer_reduce=True,
                      stream=self.side.cuda_stream if on_side else None, allow_wide=self.side is None)
        self.arena.commit(cf + splits * M * N)
        (self.red_side if on_side else self.red).append((region.data_ptr() + 4 * cf, out.data_ptr() + 4 * out_off, 0, splits, M, N, ldc,
                                                         int(bool(accumulate)), 0))
        if on_side:
            self._flush_side()

    def colsum(self, dy, out):
        """out[cols] = column sums of dy, which nobody writes any more"""
        _chk(dy, out)
        assert dy.is_contiguous()
        row = (dy.data_ptr(), out.data_ptr(), 0, dy.shape[0], dy.shape[1], dy.shape[1], 0)
        self.keep.append(dy)               # read when its batch goes out
        if self.side is None:
            self.col.append(row)
            return
        if self.synced is not dy:
            self._side_follows_main()
            self.synced = dy
        self.col_side.append(row)
        self._flush_side()

    def _table(self, name, rows, fmt):
        """device copy of an item table; (tensor, True when it was uploaded just now -- on the CURRENT stream)"""
        import struct
        key = tuple(rows)
        cached = self.state['tables'].get(name)
        if cached is not None and cached[0] == key:
            return cached[1], False
        raw = b''.join(struct.pack(fmt, *r) for r in rows)
        DeferredGrads.uploads += 1          # (diagnostic: tables re-uploaded because the item list changed)
        t = upload_bytes(raw, self.dev)
        self.state['tables'][name] = (key, t)
        return t, True

    def _flush_side(self, force=False):
        if len(self.red_side) + len(self.col_side) < (1 if force else self.SIDE_BATCH):
            return
        st = self.side.cuda_stream
        for kind, rows, fmt, fn in (('red', self.red_side, '<QQQiiiiii', lib.capmi_splitk_reduce_batch),
                                    ('col', self.col_side, '<QQQiiii', lib.capmi_colsum_batch)):
            if rows:
                t, uploaded = self._table('%s_side%d' % (kind, self.side_batches), rows, fmt)
                if uploaded:
                    self._side_follows_main()         # the table's copy was enqueued on the main stream
                check(fn(t.data_ptr(), len(rows), st), 'capmi_%s_batch (side stream)' % kind)
        self.red_side, self.col_side = [], []
        self.side_batches += 1

    def flush(self):
        if self.side is not None and self.ev_used:
            self._flush_side(force=True)
            torch.cuda.current_stream().wait_stream(self.side)
        if self.group:
            gemm_group_tn(self.group, cache_key=('deferred', str(self.dev)))
            self.group = []
        if self.red:
            t, _ = self._table('red', self.red, '<QQQiiiiii')
            check(lib.capmi_splitk_reduce_batch(t.data_ptr(), len(self.red), stream_ptr()), 'capmi_splitk_reduce_batch')
        if self.col:
            t, _ = self._table('col', self.col, '<QQQiiii')
            check(lib.capmi_colsum_batch(t.data_ptr(), len(self.col), stream_ptr()), 'capmi_colsum_batch')
        self.red, self.col, self.keep = [], [], []

    def abandon(self):
        """the backward raised: nothing is finished, but GEMMs already enqueued on the side stream may still be reading the dy / x
        tensors in `keep`.  The main stream must wait for them before those blocks return to the caching allocator (a caller
        that catches the error -- an OOM retry -- would otherwise be handed memory a side-stream kernel still reads)."""
        if self.side is not None and self.ev_used:
            torch.cuda.current_stream().wait_stream(self.side)
        self.red, self.col, self.keep = [], [], []
        self.red_side, self.col_side = [], []
        if self.group is not None:
            self.group = []


def linear(x, weight, bias=None, relu=False, mul_mask=None, row_div=1, rows=None, ws=None, out=None):
    """y = act(x @ weight.T + bias) (* mask); x [M0,K] read as row r -> r // row_div.  out: optional [M,N] contiguous target."""
    _chk(x, weight, bias, mul_mask, out)
    M = x.shape[0] * row_div if rows is None else rows
    N, K = weight.shape
    if out is None:
        out = torch.empty(M, N, dtype=_f32, device=x.device)
    gemm([(x, x.shape[1], weight, K, K, row_div)], M, N, out, bias=bias, relu=relu, mul_mask=mul_mask, ws=ws)
    return out


def matmul_nn(a, b, out=None, ws=None):
    """a [M,K] @ b [K,N]"""
    _chk(a, b)
    M, K = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=_f32, device=a.device)
    gemm([(a, K, b, N, K, 1)], M, N, out, a_layout=0, b_layout=1, ws=ws)
    return out


def matmul_tn(a, b, out=None, ws=None):
    """a [K,M]^T @ b [K,N]  (weight gradients dW = dY^T X)"""
    _chk(a, b)
    K, M = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=_f32, device=a.device)
    gemm([(a, M, b, N, K, 1)], M, N, out, a_layout=1, b_layout=1, ws=ws)
    return out


def attention_fwd(att_h, p_att, att, mask, w, b, n, row_img=None):
    _chk(att_h, p_att, att, mask, w, b)
    B, K, A = p_att.shape
    R = att.shape[2]
    N = att_h.shape[0]
    assert row_img is not None or N == B * n
    ctx = torch.empty(N, R, dtype=_f32, device=att.device)
    alpha = torch.empty(N, K, dtype=_f32, device=att.device)
    check(lib.capmi_attention_fwd(ptr(att_h), ptr(p_att), ptr(att), ptr(mask), ptr(w), ptr(b), ptr(ctx), ptr(alpha),
                                  B, n, K, A, R, ptr(row_img), N, stream_ptr()), 'capmi_attention_fwd')
    return ctx, alpha


def attention_bwd(d_ctx, att_h, alpha, p_att, att, mask, w, n, row_img=None):
    _chk(d_ctx, att_h, alpha, p_att, att, mask, w)
    B, K, A = p_att.shape
    R = att.shape[2]
    N = att_h.shape[0]
    d_att_h = torch.empty(N, A, dtype=_f32, device=att.device)
    d_e = torch.empty(N, K, dtype=_f32, device=att.device)
    check(lib.capmi_attention_bwd(ptr(d_ctx), d_ctx.shape[1], ptr(att_h), ptr(alpha), ptr(p_att), ptr(att), ptr(mask),
                                  ptr(w), ptr(d_att_h), ptr(d_e), B, n, K, A, R, ptr(row_img), N, stream_ptr()),
          'capmi_attention_bwd')
    return d_att_h, d_e


def attention_bwd_batched(d_ctx_all, att_h_all, alpha_all, d_e_all, p_att, w, n, R):
    _chk(d_ctx_all, att_h_all, alpha_all, d_e_all, p_att, w)
    T = att_h_all.shape[0]
    B, K, A = p_att.shape
    dev = p_att.device
    d_att = torch.empty(B, K, R, dtype=_f32, device=dev)
    d_p_att = torch.empty(B, K, A, dtype=_f32, device=dev)
    d_w = torch.empty(A, dtype=_f32, device=dev)
    d_b = torch.empty(1, dtype=_f32, device=dev)
    check(lib.capmi_attention_bwd_batched(ptr(d_ctx_all), d_ctx_all.shape[-1], ptr(att_h_all), ptr(alpha_all),
                                          ptr(d_e_all), ptr(p_att), ptr(w), ptr(d_att), ptr(d_p_att), ptr(d_w), ptr(d_b),
                                          T, B, n, att_h_all.shape[1], K, A, R, stream_ptr()), 'capmi_attention_bwd_batched')
    return d_att, d_p_att, d_w, d_b


def lstm_cell_fwd(partial, splits, b_ih, b_hh, c_prev, row_bias=None, row_bias_div=1, out_mask=None, want_drop=False):
    N, R = c_prev.shape
    dev = c_prev.device
    h = torch.empty(N, R, dtype=_f32, device=dev)
    c = torch.empty(N, R, dtype=_f32, device=dev)
    gates = torch.empty(N, 4 * R, dtype=_f32, device=dev)
    h_drop = torch.empty(N, R, dtype=_f32, device=dev) if (want_drop or out_mask is not None) else None
    check(lib.capmi_lstm_cell_fwd(ptr(partial), splits, ptr(b_ih), ptr(b_hh), ptr(row_bias), row_bias_div, None, ptr(c_prev),
                                  ptr(h), ptr(c), ptr(gates), ptr(out_mask), ptr(h_drop), N, R, stream_ptr()),
          'capmi_lstm_cell_fwd')
    return h, c, gates, h_drop


def lstm_cell_bwd(dh, dc_next, gates, c_prev, c_new, dh_mask=None, dh_b=None, dh_c=None):
    N, R = c_prev.shape
    dev = c_prev.device
    dg = torch.empty(N, 4 * R, dtype=_f32, device=dev)
    dc_prev = torch.empty(N, R, dtype=_f32, device=dev)
    check(lib.capmi_lstm_cell_bwd(ptr(dh), R, ptr(dh_mask), ptr(dh_b), R, ptr(dh_c), R, ptr(dc_next), ptr(gates),
                                  ptr(c_prev), ptr(c_new), ptr(dg), ptr(dc_prev), N, R, stream_ptr()),
          'capmi_lstm_cell_bwd')
    return dg, dc_prev


def embed_fwd(it, E, mask=None, relu=True):
    N = it.shape[0]
    x = torch.empty(N, E.shape[1], dtype=_f32, device=E.device)
    check(lib.capmi_embed_fwd(ptr(it), 1, None, ptr(E), ptr(mask), ptr(x), N, E.shape[1], int(relu), stream_ptr()),
          'capmi_embed_fwd')
    return x


def _keep_scale(p):
    """1 / (1 - p) rounded as the mask kernels round it (float32 throughout)"""
    import numpy as np
    return float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def dropout_mask(shape, p, seed, offset, device):
    m = torch.empty(shape, dtype=_f32, device=device)
    check(lib.capmi_dropout_mask(ptr(m), m.numel(), float(p), int(seed), int(offset), stream_ptr()), 'capmi_dropout_mask')
    m._capmi_scale = (_keep_scale(p), m._version)       # (scale, tensor version: an in-place edit by a caller voids it)
    return m


def dropout_masks(specs, p, seed):
    """All dropout masks of a rollout in one launch.  specs: list of (shape, philox_offset, keep_from) with keep_from = None or
    the first row index (of the second-to-last dimension) that stays in eval mode (mask 1.0).  Returns the mask tensors."""
    dev = specs[0][3] if len(specs[0]) > 3 else None
    descs = (_lib.MaskDesc * len(specs))()
    outs = []
    for i, sp in enumerate(specs):
        shape, offset, keep_from = sp[0], sp[1], sp[2]
        m = torch.empty(shape, dtype=_f32, device=sp[3])
        if sp[2] is None:                       # (eval-mode rows hold 1.0: not a pure keep-scale mask)
            m._capmi_scale = (_keep_scale(p), m._version)
        outs.append(m)
        descs[i].mask, descs[i].count, descs[i].offset = m.data_ptr(), m.numel(), int(offset)
        descs[i].row_len = int(shape[-1])
        descs[i].rows = int(shape[-2]) if len(shape) >= 2 else 1
        descs[i].keep_from = -1 if keep_from is None else int(keep_from)
    check(lib.capmi_dropout_masks(descs, len(specs), float(p), int(seed), stream_ptr()), 'capmi_dropout_masks')
    return outs


def reward_criterion(sel, seq, reward, n_used, per_row=False):
    """capmi_reward_criterion: sel [N_all, L] f32, seq [N_all, L] int64 (rows 0..n_used-1 are scored), reward [n_used] or
    [n_used, L] f32.  Returns (loss [1] or [n_used], gcoef [N_all, L])."""
    _chk(sel, seq)
    if not (reward.is_cuda and reward.dtype == _f32):
        raise _lib.CapmiError('reward must be a float32 device tensor')
    N_all, L = sel.shape
    loss = torch.empty(n_used if per_row else 1, dtype=_f32, device=sel.device)
    gcoef = torch.empty(N_all, L, dtype=_f32, device=sel.device)
    rs, cs = (reward.stride(0), reward.stride(1)) if reward.ndim == 2 else (reward.stride(0), 0)
    check(lib.capmi_reward_criterion(ptr(sel), sel.stride(0), ptr(seq), seq.stride(0), ptr(reward), rs, cs, int(n_used), N_all, L,
                                     int(per_row), ptr(loss), ptr(gcoef), stream_ptr()), 'capmi_reward_criterion')
    return loss, gcoef


def colsum(x, out=None, accumulate=False):
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, dtype=_f32, device=x.device)
    check(lib.capmi_colsum(ptr(x), rows, cols, cols, ptr(out), int(accumulate), stream_ptr()), 'capmi_colsum')
    return out


def relu_mask_bwd(dy, y_ref, mask):
    """dx = dy * mask * (y_ref > 0).  When y_ref is the output AFTER the mask and the mask came from dropout_mask(s) (it carries its
    keep-scale 1 / (1 - p) as `_capmi_scale`), the mask is not read at all: y_ref > 0 says where it kept the element."""
    dx = torch.empty_like(dy)
    tag = getattr(mask, '_capmi_scale', None) if (mask is not None and y_ref is not None) else None
    # (trusted only while nobody has written the mask in place since the mask kernel made it: tests inject masks with copy_)
    scale = tag[0] if (tag is not None and tag[1] == mask._version) else None
    if scale is not None and dy.numel() % 4 == 0 and (dy.data_ptr() | y_ref.data_ptr() | dx.data_ptr()) % 16 == 0 \
            and os.environ.get('CAPMI_RELU_SCALE', '1') != '0':
        check(lib.capmi_relu_scale_bwd(ptr(dy), ptr(y_ref), float(scale), ptr(dx), dy.numel(), stream_ptr()), 'capmi_relu_scale_bwd')
        return dx
    check(lib.capmi_relu_mask_bwd(ptr(dy), ptr(y_ref), ptr(mask), ptr(dx), dy.numel(), stream_ptr()), 'capmi_relu_mask_bwd')
    return dx


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, clip, grad_scale, step):
    check(lib.capmi_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, clip,
                              grad_scale, step, stream_ptr()), 'capmi_adam_step')


class StepState:
    """The per-iteration record of a graph-capturable training step in device memory (capmi.h capmi_step_state): the epoch of the
    random streams, Adam's step count with its bias corrections, the learning rate.  `advance()` is the first launch of every
    iteration (stepped or captured); while `bound()` every dropout / sampling kernel offsets its seed argument by the epoch."""

    def __init__(self, device, adam_step=0, epoch=0, lr=0.0):
        import numpy as np
        self.buf = torch.zeros(32, dtype=torch.uint8, device=device)
        host = _lib.StepState(epoch=int(epoch), adam_step=int(adam_step), lr=float(lr), bc1=1.0, bc2_sqrt=1.0)
        self.buf.copy_(torch.from_numpy(np.frombuffer(bytes(host), dtype=np.uint8).copy()))
        self.lr = float(lr)

    def advance(self, beta1, beta2):
        check(lib.capmi_step_advance(self.buf.data_ptr(), beta1, beta2, stream_ptr()), 'capmi_step_advance')

    def set_lr(self, lr):
        """outside the captured graph; a launch only when the schedule moved the rate (misc.py set_lr)"""
        if float(lr) != self.lr:
            check(lib.capmi_step_set_lr(self.buf.data_ptr(), float(lr), stream_ptr()), 'capmi_step_set_lr')
            self.lr = float(lr)

    def read(self):
        """host copy (a synchronisation: tests / checkpoints only)"""
        return _lib.StepState.from_buffer_copy(bytes(self.buf.cpu().numpy().tobytes()))

    class _Bound:
        def __init__(self, st):
            self.st = st

        def __enter__(self):
            self.prev = C.c_void_p()
            check(lib.capmi_rng_bind_epoch(self.st.buf.data_ptr(), C.byref(self.prev)), 'capmi_rng_bind_epoch')

        def __exit__(self, *a):
            check(lib.capmi_rng_bind_epoch(self.prev.value, None), 'capmi_rng_bind_epoch')
            return False

    def bound(self):
        return StepState._Bound(self)


def adam_step_dyn(p, g, m, v, state, beta1, beta2, eps, weight_decay, clip, grad_scale):
    check(lib.capmi_adam_step_dyn(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), state.buf.data_ptr(), beta1, beta2, eps, weight_decay,
                                  clip, grad_scale, stream_ptr()), 'capmi_adam_step_dyn')


def logsoftmax_bwd(g_dense, sparse, seq_logp, live, dlogits, N, L, T, V1, raw=False):
    """d(logits) [T,N,V1] from the loss gradient w.r.t. the dense log-probs: dense (`g_dense` [N,L,V1]), sparse
    (`sparse`, a _lib.SparseLogpGrad from sparse_logp.split_grad) or both.
    raw: the rollout stored the LOGITS (CAPMI_SELECT_RAW; AttModel._sample(output_logsoftmax=0), AttModel.py:171-175, 265) -- the loss
    gradient IS d(logits) on the live rows, no softmax Jacobian (capmi_sparse_logp_grad.raw)."""
    if raw:
        sp = _lib.SparseLogpGrad() if sparse is None else _lib.SparseLogpGrad.from_buffer_copy(sparse)
        sp.raw = 1
        sparse = sp
    if sparse is not None:
        check(lib.capmi_logsoftmax_bwd_sparse(C.byref(sparse), ptr(g_dense), ptr(seq_logp), ptr(live), ptr(dlogits), N, L, T, V1,
                                              stream_ptr()), 'capmi_logsoftmax_bwd_sparse')
    else:
        check(lib.capmi_logsoftmax_bwd(ptr(g_dense), ptr(seq_logp), ptr(live), ptr(dlogits), N, L, T, V1, stream_ptr()),
              'capmi_logsoftmax_bwd')


def logsoftmax_select(logits, step, L, mode, temperature, gumbel, seed, forced, no_finish_mask, seq, it_next, unfinished,
                      seq_logp, sel_logp, live, top_k=0, top_p=0.0, splits=1, stride=0, bias=None, shape=None, next_embed=None):
    """capmi_logsoftmax_select_partial with the optional top-k / nucleus filter; used by the host-stepped decoders
    (Transformer, AoA).  logits: finished [N,V1] (one slab, no bias), or -- with shape=(N,V1) -- the `splits` K-slice slabs a
    deferred logit GEMM left `stride` floats apart, finished here together with `bias`.
    next_embed: dict(E, mask, x, it_save, relu, x_planes) -- the workgroup that chose a row's token also writes the NEXT step's
    input embedding x[r] = relu?(E[token]) * mask[r] (+ its planes) and it_save[r] = token (capmi_next_embed)."""
    N, V1 = logits.shape if shape is None else shape
    flt = _lib.SampleFilter(int(top_k), float(top_p))
    ne = None
    if next_embed is not None:
        ne = _lib.NextEmbed(ptr(next_embed['E']), ptr(next_embed.get('mask')), ptr(next_embed['x']), ptr(next_embed.get('it_save')),
                            int(next_embed['E'].shape[1]), int(next_embed.get('relu', 1)), ptr(next_embed.get('x_planes')), None)
    check(lib.capmi_logsoftmax_select_partial(ptr(logits), int(splits), int(stride), ptr(bias), N, V1, step, L, mode, None,
                                              float(temperature),
                                              ptr(gumbel), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(forced),
                                              0 if forced is None else forced.shape[1], int(no_finish_mask), ptr(seq), L,
                                              ptr(it_next), ptr(unfinished), ptr(seq_logp), ptr(sel_logp), ptr(live),
                                              None if ne is None else C.byref(ne),
                                              C.byref(flt) if (top_k or top_p) else None, stream_ptr()),
          'capmi_logsoftmax_select_partial')


def caption_stats(seq_logp, seq):
    """(entropy [N], perplexity [N]) of a decode (eval_utils.py:173-174) from its dense log-probs [N, L, V1] and tokens [N, L]: one
    pass over seq_logp, nothing dense is built (capmi_caption_stats)."""
    _chk(seq_logp, seq)
    N, L, V1 = seq_logp.shape
    if seq.dtype != torch.int64 or tuple(seq.shape) != (N, L) or seq_logp.dtype != _f32:
        raise _lib.CapmiError('caption_stats: seq_logp [N,L,V1] float32 and seq [N,L] int64 expected')
    scratch = torch.empty(2 * N * L, dtype=_f32, device=seq_logp.device)
    ent = torch.empty(N, dtype=_f32, device=seq_logp.device)
    ppl = torch.empty(N, dtype=_f32, device=seq_logp.device)
    check(lib.capmi_caption_stats(ptr(seq_logp), ptr(seq), N, L, V1, ptr(scratch), ptr(ent), ptr(ppl), stream_ptr()), 'capmi_caption_stats')
    return ent, ppl


def clip_len(att_masks, width=None):
    """Longest valid region count of the batch -- the K that clip_att (AttModel.py:106-112) truncates to.

    ``int(mask.sum(1).max())`` is a blocking device->host round trip; the value is cached on the mask tensor
    (``_capmi_kmax``) so one step pays it at most once, and the loaders / DevicePrefetcher stamp it from the host-side
    batch (they built the mask, they know kmax), so a training step never syncs for it."""
    if att_masks is None:
        return width
    k = getattr(att_masks, '_capmi_kmax', None)
    if k is None:
        k = int(att_masks.long().sum(1).max())
        try:
            att_masks._capmi_kmax = k
        except Exception:
            pass
    return k
