#!/usr/bin/env python3
"""bench.py -- captions/sec of the UpDown SCST training step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full SCST iteration per GPU on a synthetic batch (configs[2]: bs 10 x train_sample_n 5,
36x2048 region features, seq_len 20, vocab 9487): references packed + cooked for CIDEr-D (on the prefetcher's
copy stream) + greedy rollout + sampled rollout (dropout on) + CIDEr-D reward + RewardCriterion + BPTT +
(1 RCCL all-reduce of the flat gradient if N > 1) + value clip 0.1 + Adam.  The steps ROTATE over 4 distinct
batches whose features are resident in HBM before the timed region (round 3; one batch, references cooked once
outside the loop before).  Weak scaling: every rank runs its own 10 images.  Prints ONE JSON line (rank 0).

--config selects the other BASELINE.json configurations with the same JSON schema (their own `metric` name):
updown_xe (configs[1], bs64), transformer_xe (configs[3], bs64), aoa_nsc (configs[4], bs10 x 5), newfc_xe (configs[0]).

Extra objects on the line:
  roofline      dominant kernel (skinny weight-streaming MFMA GEMM of the decode step): algorithmic bytes
                / average launch duration from HIP events recorded inside the timed region
  attention     the same for the fused region-attention kernel named by north_star
  cpu_baseline  the oracle's SCST iteration (oracle/scst_step.py, kind "port") on this box's host cores
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC (RCCL needs it)

import torch                                                # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DF_IMAGES = 10000
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA (v_mfma_f32_32x32x2_f32), 155 TF measured
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (32x32x16), 2495 TF measured


def measured_copy_gbs(dev, mb=1024, iters=8):
    """second denominator: what a plain device-to-device copy moves on THIS box, read + write bytes."""
    src = torch.empty(mb << 18, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return 2.0 * src.numel() * 4 * iters / (a.elapsed_time(b) * 1e-3) / 1e9


def attention_large_batch(dev, B=1024, sets=6, iters=30):
    """The fused region-attention kernel at an evaluation-sized batch (B images x 36 regions, one caption row each), where
    a launch moves enough unique bytes (229 MB) to be bandwidth-bound: the north_star's HBM-roofline target applies here;
    at the bs10 SCST shape a launch moves 2-5 MB and is latency-bound.  The launches ROTATE over `sets` independent input
    sets (6 x 229 MB = 1.4 GB > 5 x the 256 MiB Infinity Cache), so every launch streams from HBM; the same-buffer figure
    (inputs resident in the Infinity Cache) is reported beside it as `cached_gbs`.  Kernel time from the in-dispatch HIP
    events of libcapmi (class 3), i.e. without host launch overhead."""
    import ctypes as C
    from imagecaptioning.pytorch_amd import ops, _lib
    lib = _lib.lib
    K, A, R = 36, 512, 1000
    w = torch.randn(A, device=dev) * 0.1
    bb = torch.zeros(1, device=dev)
    data = [(torch.randn(B, A, device=dev), torch.randn(B, K, A, device=dev), torch.randn(B, K, R, device=dev)) for _ in range(sets)]

    def run(rotate):
        for i in range(sets):
            ops.attention_fwd(*data[i if rotate else 0], None, w, bb, 1)
        torch.cuda.synchronize()
        lib.capmi_prof_reset()
        lib.capmi_prof_enable(1 << 3)
        for i in range(iters):
            ops.attention_fwd(*data[i % sets if rotate else 0], None, w, bb, 1)
        torch.cuda.synchronize()
        lib.capmi_prof_enable(0)
        ms, n, b_, f_ = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        assert lib.capmi_prof_read(3, C.byref(ms), C.byref(n), C.byref(b_), C.byref(f_)) == 0 and n.value == iters
        lib.capmi_prof_reset()
        return ms.value / iters * 1e3
    us_hbm, us_cached = run(True), run(False)
    byts = 4.0 * (B * K * (A + R) + B * (A + R + K))
    return {'B': B, 'n': 1, 'rotating_sets': sets, 'working_set_mb': round(sets * byts / 1e6), 'avg_launch_us': round(us_hbm, 1),
            'unique_bytes_per_launch': round(byts), 'achieved_gbs': round(byts / us_hbm / 1e3, 1),
            'frac_of_8tbs': round(byts / us_hbm / 1e3 / HBM_PEAK_GBS, 4),
            'frac_of_achievable_6290gbs': round(byts / us_hbm / 1e3 / 6290.0, 4),    # MI355X_MICROARCH.md: achievable HBM rate
            'cached_avg_launch_us': round(us_cached, 1), 'cached_gbs': round(byts / us_cached / 1e3, 1)}


def pmc_traffic(instance=False):
    """HBM bytes per decode-GEMM launch from the PMC passes of this same command (FETCH_SIZE and WRITE_SIZE need
    separate rocprofv3 --pmc runs, so they cannot be sampled inside this process): profiles/r0N_pmc_traffic.json,
    written by scripts/tools_pmc_traffic.py with the gfx950 corrections of MI355X_MICROARCH.md.  None if absent.
    The line says so itself (`roofline.traffic_source`): imported, not measured in this run."""
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        with open(os.path.join(here, 'profiles', PMC_FILE)) as f:
            d = json.load(f)
        if instance:       # the dominant kernel alone (the weight-streaming launches of gemm_lc_kernel<true,2>: LSTM gates, logit)
            for k, v in d['kernels'].items():
                if 'gemm_lc_kernel<true, 2' in k and '[stream]' in k:
                    return round(v['fetch_bytes_corrected'] + v['write_bytes'])
        return round(d['decode_gemm']['traffic_bytes'])
    except (OSError, KeyError, ValueError):
        return None


PMC_FILE = 'r06_pmc_traffic.json'


def prof_read(lib, cls):
    ms, n, b, f = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    rc = lib.capmi_prof_read(cls, C.byref(ms), C.byref(n), C.byref(b), C.byref(f))
    assert rc == 0, rc
    return ms.value, n.value, b.value, f.value


def _opt(name):
    from imagecaptioning.pytorch_amd import synthetic
    if name == 'updown':
        return synthetic.updown_opt()
    if name == 'newfc':           # configs/fc.yml: opts.py defaults rnn_size = input_encoding_size = 512
        return synthetic.updown_opt(caption_model='newfc', input_encoding_size=512, rnn_size=512)
    if name == 'transformer':     # configs/transformer/transformer.yml:23-28
        return synthetic.updown_opt(caption_model='transformer', input_encoding_size=512, rnn_size=2048, d_model=512, d_ff=2048,
                                    N_enc=6, N_dec=6, num_att_heads=8, dropout=0.1, drop_prob_lm=0.5)
    if name == 'aoa_nsc':         # configs/aoa.yml + aoa_nsc.yml
        return synthetic.updown_opt(caption_model='aoa', input_encoding_size=1024, rnn_size=1024, att_hid_size=512, num_heads=8,
                                    multi_head_scale=1, use_multi_head=2, refine=1, refine_aoa=1, use_ff=0, decoder_type='AoA',
                                    mean_feats=1, ctx_drop=1, dropout_aoa=0.3, drop_prob_lm=0.5, train_sample_n=5,
                                    structure_loss_type='new_self_critical', structure_loss_weight=1.0, label_smoothing=0.2,
                                    learning_rate=2e-5)
    raise KeyError(name)


# name -> (opt, per-GPU batch, (sc_flag, struc_flag), BASELINE.json configs[] index, workload text)
CONFIGS = {
    'updown_scst': ('updown', 10, (True, False), 2,
                    'UpDown SCST (BASELINE configs[2]): per-GPU batch 10 x train_sample_n 5, 36x2048 bottom-up feats, R=E=1000 A=512, '
                    'vocab 9487, seq_len 20, references packed+cooked + greedy baseline + CIDEr-D + RewardCriterion + BPTT + clip 0.1 + Adam'),
    'updown_xe': ('updown', 64, (False, False), 1,
                  'UpDown XE (BASELINE configs[1]): per-GPU batch 64 x 5 captions, 36x2048 feats, seq_len 20 (T = 21 teacher-forced steps), '
                  'LanguageModelCriterion + BPTT + clip 0.1 + Adam'),
    'transformer_xe': ('transformer', 64, (False, False), 3,
                       'Transformer XE (BASELINE configs[3]): per-GPU batch 64 x 5 captions, 36x2048 feats, d=512 d_ff=2048 h=8 N=6, '
                       'seq_len 20, LanguageModelCriterion + backward + clip 0.1 + Adam'),
    'aoa_nsc': ('aoa_nsc', 10, (False, True), 4,
                'AoA new-self-critical (BASELINE configs[4]): per-GPU batch 10 x train_sample_n 5, R=E=1024 h=8, 6 refiner layers, '
                'sampled rollouts + CIDEr-D + StructureLosses(new_self_critical) + BPTT + clip 0.1 + Adam'),
    'newfc_xe': ('newfc', 10, (False, False), 0,
                 'NewFC XE (BASELINE configs[0], configs/fc.yml): batch 10 x 5 captions, 2048-d fc feats, R=E=512, seq_len 20, '
                 'LanguageModelCriterion + BPTT + clip 0.1 + Adam'),
}


class _Rotating:
    """The reference's loader contract (dataloader.py:229-258 batch dict) over a few distinct synthetic batches whose features
    already live in HBM; `gts` stay host arrays so that the prefetcher packs + cooks them per batch like for real data."""

    def __init__(self, items):
        self.items, self.i = items, 0

    def get_batch(self, split):
        d = dict(self.items[self.i % len(self.items)])
        self.i += 1
        return d


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: re-run this command under `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>` (the driver's documented form)
    and hand its output through -- rank 0 still prints the ONE JSON line.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '8')      # (torchrun would set 1 and warn)
    print('bench.py: launching %d ranks: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


OTHER_CONFIGS = ('updown_xe', 'transformer_xe', 'aoa_nsc', 'newfc_xe')


def other_configs(budget_s=150.0):
    """The four BASELINE configurations the headline is NOT quoted on, each as a short pass of this same script in a child process
    (`--config X --brief`: 8 init + 3 warm-up + 8 timed steps, its own CPU baseline on a smaller sample), AFTER the headline
    measurement is complete; a child that fails or exceeds its budget becomes {'error': ...} and never costs the headline."""
    import subprocess
    out = {}
    # (label, config, extra environment).  The last entry is NOT a BASELINE configuration: Transformer XE with one encoder pass per
    # image in train mode (tie_encoder_dropout; the reference encodes every caption row under its own dropout masks, which is the
    # default here) -- reported beside the faithful line because r5's Transformer numbers were measured that way
    runs = [(n, n, {}) for n in OTHER_CONFIGS] + [('transformer_xe_tied_encoder', 'transformer_xe', {'CAPMI_TIE_ENC': '1'})]
    for name, cfg_name, extra_env in runs:
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--config', cfg_name, '--steps', '8', '--warmup', '3', '--brief']
        if extra_env:
            cmd.append('--no-cpu-baseline')
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget_s, text=True,
                               env=dict(os.environ, **extra_env))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not lines:
                out[name] = {'error': 'rc %d: %s' % (r.returncode, (r.stderr or '').strip().splitlines()[-1:] or '')}
                continue
            d = json.loads(lines[-1])
            rf, cb = d.get('roofline') or {}, d.get('cpu_baseline') or {}
            out[name] = {'metric': d['metric'], 'baseline_config': d['config']['baseline_config'], 'ms_per_step': d['ms_per_step'],
                         'step_ms': d.get('step_ms'),
                         'captions_per_s': d['value'], 'steps': d['steps'], 'warmup': d['warmup'], 'dtype': d['dtype'],
                         'roofline': {k: rf.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us')},
                         'cpu_baseline': {k: cb.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample')} if cb else None,
                         'wall_s': round(time.perf_counter() - t0, 1)}
            if extra_env:
                out[name]['switches'] = extra_env
                out[name]['note'] = ('not the reference dataflow: opt.tie_encoder_dropout -- the n caption rows of an image share their '
                                     'encoder dropout masks (one encoder pass per image instead of per caption row)')
        except subprocess.TimeoutExpired:
            out[name] = {'error': 'exceeded its %.0f s budget' % budget_s}
        except Exception as e:                      # noqa: BLE001 -- nothing here may cost the headline line
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def ddp_other_configs(dev, world, rank, local, dist, names=('transformer_xe', 'aoa_nsc'), steps=6, warmup=3, df_ref=None):
    """N > 1 only, AFTER the headline line and the mode sweep: the two configurations BASELINE.json runs as 8-GPU data parallel
    (configs[3] Transformer XE, configs[4] AoA new-self-critical; reference tools/train_pl.py:459-480) for a few steps on the SAME
    process group -- graph_step.TrainStep (captured forward + backward, then the ONE flat RCCL all-reduce and clip + Adam), barrier +
    synchronize around the timed steps, MAX over ranks.  Returns {name: {...}} on every rank (rank 0 puts it on the line)."""
    from imagecaptioning.pytorch_amd import synthetic
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper
    from imagecaptioning.pytorch_amd.captioning.utils import rewards
    from imagecaptioning.pytorch_amd.graph_step import TrainStep

    def sync():
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()
        torch.cuda.synchronize()

    out = {}
    for name in names:
        cfg = CONFIGS[name]
        try:
            opt = _opt(cfg[0])
            sc_flag, struc_flag = cfg[2]
            torch.manual_seed(1234)
            model = models.setup(opt).to(dev)
            flat = model.flatten_parameters_()
            lw = LossWrapper(model, opt)
            B, n, L = cfg[1], opt.train_sample_n, opt.max_length
            rewards.reset_scorer()
            if sc_flag or struc_flag:
                if df_ref is None:
                    df_ref = synthetic.document_frequency(synthetic.corpus(DF_IMAGES, seed=7))
                rewards.init_scorer(df_ref, device=dev)
            batches = []
            for b in range(2):
                f_, a_ = synthetic.batch(B, seed=4321 + 1000 * b + rank, device=dev)
                lab = msk = None
                if not (sc_flag or struc_flag):
                    lab, msk = synthetic.xe_labels(B, n=5, L=L, seed=4321 + 1000 * b + rank)
                    lab, msk = lab.to(dev), msk.to(dev)
                batches.append({'fc_feats': f_, 'att_feats': a_, 'att_masks': None, 'labels': lab, 'masks': msk,
                                'gts': synthetic.corpus(B, seed=300 + 10 * b + rank)})
            ar = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

            def all_reduce():
                ar[0].record()
                scale = flat.all_reduce()
                ar[1].record()
                return scale
            ts = TrainStep(lw, flat, opt, dev, world=world, all_reduce=all_reduce)
            for i in range(warmup + 2):                 # (+2: the first batch of a shape steps, the second is captured)
                loss, _ = ts(batches[i % 2], sc_flag, struc_flag)
            sync()
            t0 = time.perf_counter()
            ar_ms = []
            for i in range(steps):
                loss, _ = ts(batches[i % 2], sc_flag, struc_flag)
                if i == steps - 1:
                    torch.cuda.synchronize()
                    ar_ms.append(ar[0].elapsed_time(ar[1]))
            sync()
            d = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
            dt = float(d.item())
            out[name] = {'baseline_config': 'BASELINE.json configs[%d]' % cfg[3], 'n_gpus': world, 'steps': steps, 'warmup': warmup,
                         'ms_per_step': round(dt / steps * 1e3, 3), 'captions_per_s': round(B * n * world * steps / dt, 1),
                         'global_batch': B * world, 'loss': float(loss.detach()),
                         'step_issue': 'hipGraph replay up to the flat gradient, then all-reduce + clip/Adam' if ts.replays else 'stepped',
                         'collective': {'backend': dist.get_backend(), 'mode': 'one flat all-reduce per step',
                                        'bytes': int(flat.grad.numel() * 4), 'allreduce_ms': round(ar_ms[-1], 3) if ar_ms else None}}
            del ts, lw, flat, model, batches
            torch.cuda.empty_cache()
        except Exception as e:                      # noqa: BLE001 -- nothing here may cost the headline line
            out[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=10, help='per-GPU images (weak scaling, the default)')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling (SURVEY 8d: global bs 80): total images per step, split evenly over the ranks')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='do not bracket kernels with HIP events (for rocprofv3 runs)')
    ap.add_argument('--cpu-iters', type=int, default=5)
    ap.add_argument('--config', default='updown_scst', choices=sorted(CONFIGS),
                    help='BASELINE.json configuration (default: configs[2], the one the headline metric is quoted on)')
    ap.add_argument('--brief', action='store_true',
                    help='the headline objects only: no large-batch attention pass, no copy measurement, no early-exit line, no other '
                         'configurations, a 1-iteration CPU sample (how the default run times the other BASELINE configurations)')
    ap.add_argument('--no-other-configs', action='store_true', help='do not append the other BASELINE configurations to the line')
    ap.add_argument('--eos-bias', type=float, default=0.0,
                    help='add this to logit.bias[EOS]: a model that ends its captions (random weights never do), for the early-exit line')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU, as tools/train_pl.py:470-480 asks of Lightning)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or run plain `python bench.py --gpus %d`)'
                         % (args.gpus, world, args.gpus, args.gpus))
    if os.environ.get('CAPMI_DIST_BACKEND') == 'gloo':
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    # CAPMI_BENCH_FORCE_DIST=1: take the multi-GPU code path (RCCL communicator, flat all-reduce, collective timing) with ONE
    # rank -- how the N > 1 path is exercised on a 1-GPU box (tests/test_entrypoints_gpu.py); never a measurement
    multi = world > 1 or os.environ.get('CAPMI_BENCH_FORCE_DIST') == '1'
    if multi:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # nccl == RCCL on ROCm (xGMI).  CAPMI_DIST_BACKEND=gloo only exists to exercise the multi-process path on a
        # single-GPU box (two ranks sharing cuda:0), never for measurements.
        dist.init_process_group(os.environ.get('CAPMI_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
        if dist.get_backend() == 'nccl':
            # create the RCCL communicator here, on the main thread: the first gradient bucket is otherwise reduced from
            # inside autograd's backward thread
            dist.barrier(device_ids=[local])

    from imagecaptioning.pytorch_amd import synthetic, _lib
    from imagecaptioning.pytorch_amd.captioning import models
    from imagecaptioning.pytorch_amd.captioning.modules.loss_wrapper import LossWrapper
    from imagecaptioning.pytorch_amd.captioning.utils import rewards
    lib = _lib.lib

    opt = _opt(cfg[0])
    sc_flag, struc_flag = cfg[2]
    torch.manual_seed(1234)                       # identical initial weights on every rank
    model = models.setup(opt).to(dev)
    if args.eos_bias:
        with torch.no_grad():
            model.logit.bias[0] += args.eos_bias
    flat = model.flatten_parameters_()
    # Default for N > 1: ONE all-reduce of the whole flat fp32 gradient per step (north_star / SURVEY 8e).
    # CAPMI_DDP_OVERLAP=1 opts into the bucketed variant (6 collectives launched from inside the backward as the phases finish
    # their gradients, clip+Adam pipelined behind them).
    WATCH = {'phase': 'main', 'line': None, 'mode': None}
    overlap = multi and os.environ.get('CAPMI_DDP_OVERLAP', '0') == '1'
    # CAPMI_DDP_MODE=rsag: reduce-scatter -> clip+Adam on the rank's 1/N shard -> all-gather of the parameters
    sharded = multi and not overlap and os.environ.get('CAPMI_DDP_MODE', 'allreduce') == 'rsag'
    if overlap:
        flat.begin_overlap()
    if multi:
        # a hung collective cannot be caught by try/except: a watchdog thread ends the process group with a diagnostic instead
        # of letting the driver's timeout kill an unexplained run
        import threading
        limit = float(os.environ.get('CAPMI_BENCH_WATCHDOG_S', '600'))

        def _bark():
            print('rank %d: bench.py exceeded its %.0f s watchdog (a hung RCCL collective?) in phase %r; aborting'
                  % (rank, limit, WATCH.get('phase')), file=sys.stderr, flush=True)
            if WATCH.get('phase') == 'mode sweep':
                # the K timed steps and the line are done: a mode of the optional sweep hanging must not cost the measurement
                if rank == 0 and WATCH.get('line') is not None:
                    WATCH['line']['collective']['ms_per_step_by_mode'] = 'sweep hung in mode %r' % WATCH.get('mode')
                    print(json.dumps(WATCH['line']), flush=True)
                os._exit(0)
            os._exit(3)
        watchdog = threading.Timer(limit, _bark)
        watchdog.daemon = True
        watchdog.start()
    lw = LossWrapper(model, opt)
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit('--global-batch must be divisible by the number of ranks')
        args.batch = args.global_batch // world
    if args.batch == 10 and cfg[1] != 10:
        args.batch = cfg[1]
    B, n, L = args.batch, opt.train_sample_n, opt.max_length
    rewards.reset_scorer()
    df_ref = None
    if sc_flag or struc_flag:
        corpus = synthetic.corpus(DF_IMAGES, seed=7)  # DF table: 10000 synthetic "images" x 5 refs
        df, ref_len = synthetic.document_frequency(corpus)
        rewards.init_scorer((df, ref_len), device=dev)
        df_ref = (df, ref_len)
    # NB distinct batches rotate through the timed steps.  Their features are resident in HBM before the timed region; the
    # references stay host arrays and are packed + cooked for CIDEr-D per batch by the prefetcher on its copy stream, inside the
    # timed region, exactly as captioning/data/prefetch.py does for real batches (the full iteration).
    from imagecaptioning.pytorch_amd.captioning.data.prefetch import DevicePrefetcher
    NB = 4
    items = []
    for b in range(NB):
        f_, a_ = synthetic.batch(B, seed=1234 + 1000 * b + rank, device=dev)
        if sc_flag:
            lab = msk = None
        else:
            lab, msk = synthetic.xe_labels(B, n=5, L=L, seed=1234 + 1000 * b + rank)
            lab, msk = lab.to(dev), msk.to(dev)
        items.append({'fc_feats': f_, 'att_feats': a_, 'att_masks': None, 'labels': lab, 'masks': msk,
                      'gts': synthetic.corpus(B, seed=100 + 10 * b + rank), 'infos': [],
                      'bounds': {'it_pos_now': 0, 'it_max': NB * B, 'wrapped': False}})
    pf = DevicePrefetcher(_Rotating(items), dev, depth=2)
    gt_indices = torch.arange(B)

    ar_events = None          # (start, end) events around the all-reduce of the steps that measure it

    mode_now = {'overlap': overlap, 'sharded': sharded, 'none': False}
    one = torch.ones((), dtype=torch.float32, device=dev)

    # r6: the iteration itself is graph_step.TrainStep -- the trainer's own step object (tools/train.py uses the same): stepped launch
    # by launch for UpDown / NewFC (native rollouts), captured into a hipGraph and replayed for the Transformer and AoA families
    # (CAPMI_GRAPH_STEP=0: stepped everywhere).  N > 1: the graph ends at the flat gradient; all-reduce + clip/Adam follow un-captured.
    from imagecaptioning.pytorch_amd.graph_step import TrainStep

    def _all_reduce():
        if mode_now['none']:
            return 1.0
        ev = ar_events
        if ev is not None:
            ev[0].record()
        scale = flat.all_reduce()
        if ev is not None:
            ev[1].record()
        return scale
    ts = TrainStep(lw, flat, opt, dev, world=world, all_reduce=_all_reduce if multi else None)

    def step():
        if not (mode_now['overlap'] or mode_now['sharded']):
            loss, _ = ts(pf.get_batch('train'), sc_flag, struc_flag)
            return loss
        return step_legacy()

    def step_legacy():
        """the opt-in exchange modes (bucketed overlap, reduce-scatter + sharded Adam): launch by launch, host-side Adam step count"""
        nonlocal_ar = ar_events
        overlap, sharded = mode_now['overlap'], mode_now['sharded']
        data = pf.get_batch('train')
        out = lw(data['fc_feats'], data['att_feats'], data['labels'], data['masks'], None, data['gts'], gt_indices, sc_flag,
                 struc_flag, False)
        loss = out['loss']
        if loss.dim():
            loss = loss.mean()
        flat.zero_grad()
        adam = dict(lr=opt.learning_rate, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon,
                    weight_decay=opt.weight_decay, clip_value=opt.grad_clip_value)
        loss.backward(gradient=one if loss.dim() == 0 else None)       # (a cached 1.0: autograd's ones_like is an ATen fill launch)
        flat.collect_grads()
        if overlap:
            # the backward has already launched the all-reduce of every gradient bucket it finished (logit layer before
            # the BPTT loop, LSTM weights before the attention/prefill gradients); reduce the rest and run clip+Adam
            # bucket by bucket as the collectives land
            flat.finish_overlap_and_step(**adam)
        elif sharded:
            flat.sharded_step(**adam)
        elif multi and not mode_now['none']:
            if nonlocal_ar is not None:
                nonlocal_ar[0].record()
            scale = flat.all_reduce()
            if nonlocal_ar is not None:
                nonlocal_ar[1].record()
            flat.adam_step(grad_scale=scale, **adam)
        else:
            flat.adam_step(grad_scale=1.0, **adam)
        return loss

    def sync():
        if dist is not None:
            if dist.get_backend() == 'nccl':
                dist.barrier(device_ids=[local])     # bind the barrier's collective to this rank's GPU explicitly
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # device initialisation, outside the W / K protocol and reported as `init_steps`: the first process on a fresh box pays
    # one-off costs (code-object load, allocator growth to the step's working set, clock ramp) that took up to 15 % off a
    # short measurement when only W = 1-2 warm-up steps preceded it
    INIT_STEPS = 8
    for _ in range(INIT_STEPS):
        step()
    torch.cuda.synchronize()
    if os.environ.get('CAPMI_BENCH_GC_FREEZE', '1') != '0':
        # The synthetic DF corpus (10 000 reference sets -> millions of tuples and dict entries) stays alive in this process; every
        # generational collection of the interpreter walks it.  The launch-bound configurations (AoA: 1000 launches per step issued
        # from Python) measured 16.4 ms per step with it and 11 ms without: freeze the set-up heap (a long-lived training process
        # would do the same, and a real run loads the DF table from a file instead of building it as Python objects).
        import gc
        gc.collect()
        gc.freeze()
    for _ in range(args.warmup):
        step()
    lib.capmi_prof_reset()
    # The roofline's per-launch durations come from in-dispatch HIP events (start / stop events attached to the launch itself, on
    # the stream the kernel runs on) INSIDE the timed region -- but only on a sample of its steps: bracketing all ~100 decode-GEMM
    # and attention launches of every step costs 0.48 ms per step (5.18 vs 4.70 ms measured back to back), 10 % of the metric.
    # ONE of the K timed steps (two from K = 40 up) carries the events: a whole step is every launch position of the dominant kernel
    # exactly once (41 of the streaming decode GEMMs in the SCST step), and its 0.5 ms is 0.6 % of a 20-step region (r5: two sampled
    # steps were 1.2 %: 4.16 vs 4.21 ms with and without `--no-prof`, `scripts/r5_ab12.sh`).
    # r6: INSIDE the timed region only the class of the dominant kernel carries events (41 launches of the SCST step instead of 103:
    # the sampled step costs 0.2 ms instead of 0.5); the companion objects of the line (`all_decode_gemms`, `attention`) are sampled
    # on ONE extra step right behind the region (`extra_mask`)
    xe_cfg = args.config in ('updown_xe', 'transformer_xe', 'newfc_xe')
    prof_mask = (1 << 2) if xe_cfg else (1 << 9)        # fat GEMMs (XE steps) / the weight-streaming decode GEMMs
    extra_mask = (1 << 0) | (1 << 3) | ((1 << 9) if xe_cfg else 0)      # small decode GEMMs, fused attention (+ streaming ones of an XE step)
    if args.no_prof:
        sampled = set()
    elif args.steps >= 40:
        sampled = {args.steps // 3, (2 * args.steps) // 3}
    else:
        sampled = {args.steps // 2}
    host_prof = None
    if os.environ.get('CAPMI_BENCH_CPROFILE'):      # where the HOST spends the timed steps (scripts/prof_host_top.py); perturbs the timing
        import cProfile
        host_prof = cProfile.Profile()
    sync()
    t0 = time.perf_counter()
    if host_prof is not None:
        host_prof.enable()
    # one event per step boundary (r5): the spread of the per-step times says whether a slow run is slow in every step or carries
    # a few outliers (`step_ms`: min / median / max over the timed steps, by the device's clock)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for i in range(args.steps):
        marks[i].record()
        if i in sampled:
            lib.capmi_prof_enable(prof_mask)
        loss = step()
        if i in sampled:
            lib.capmi_prof_enable(0)
    marks[args.steps].record()
    if host_prof is not None:
        host_prof.disable()
    sync()
    dt = time.perf_counter() - t0
    if host_prof is not None and rank == 0:
        import io
        import pstats
        for key in ('tottime', 'cumulative'):
            buf = io.StringIO()
            pstats.Stats(host_prof, stream=buf).sort_stats(key).print_stats(30)
            print('\n'.join(l[:160] for l in buf.getvalue().split('\n') if l.strip()), file=sys.stderr)
    lib.capmi_prof_enable(0)
    graph_sample = None
    if sampled and ts.replays and not (mode_now['overlap'] or mode_now['sharded']):
        # a replayed hipGraph's launches cannot carry per-launch events (they are baked into the graph): for the captured
        # configurations the roofline sample is ONE extra iteration issued launch by launch right behind the timed region -- the same
        # launches with the same arguments (tests/test_graph_step_gpu.py), outside the K steps so that its host-bound issue does not
        # enter ms_per_step
        lib.capmi_prof_reset()
        lib.capmi_prof_enable(prof_mask | extra_mask)
        ts(pf.get_batch('train'), sc_flag, struc_flag, force_stepped=True)
        torch.cuda.synchronize()
        lib.capmi_prof_enable(0)
        graph_sample = 'one stepped iteration right behind the timed region (the timed steps replay a hipGraph)'
    elif sampled:
        lib.capmi_prof_enable(extra_mask)               # the companion classes: one extra step, outside the K timed ones
        step()
        torch.cuda.synchronize()
        lib.capmi_prof_enable(0)
    in_order = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    per_step = sorted(in_order)
    # `first`: the step right behind the opening synchronize -- the device starts it with an empty queue, so a host-stepped
    # configuration shows its issue latency there and nowhere else
    step_ms = {'min': round(per_step[0], 3), 'median': round(per_step[len(per_step) // 2], 3), 'max': round(per_step[-1], 3),
               'first': round(in_order[0], 3)}
    n_sampled = 1 if graph_sample else max(1, len(sampled))      # steps behind the dominant class's totals
    n_extra = 1                                                   # ... behind the companion classes'
    n_g = n_extra if xe_cfg else n_sampled                        # ... behind the streaming decode GEMMs'
    allreduce_ms = None
    if dist is not None and not overlap and not sharded:
        # collective time of the single flat all-reduce: HIP events around it on 5 extra (untimed) steps
        ms = []
        for _ in range(5):
            ar_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            step()
            torch.cuda.synchronize()
            ms.append(ar_events[0].elapsed_time(ar_events[1]))
        ar_events = None
        allreduce_ms = sorted(ms)[len(ms) // 2]
    def run_mode_sweep():
        mode_ms = {}
        # One driver run decides between the gradient-exchange modes: after the measurement above (the default mode, the line's
        # `value`) the same step is timed for a few iterations in every other mode and WITHOUT any exchange -- the difference to
        # the latter is the communication a mode leaves exposed.  (The replicas drift apart in the exchange-free pass: it comes last.)
        default_name = 'overlap' if overlap else ('rsag' if sharded else 'allreduce')

        def timed(name, k=6, w=2):
            WATCH['mode'] = name
            mode_now.update(overlap=(name == 'overlap'), sharded=(name == 'rsag'), none=(name == 'none'))
            if name == 'overlap':
                flat.begin_overlap()
            else:
                flat.on_grads_ready = None
            for _ in range(w):
                step()
            sync()
            t0_ = time.perf_counter()
            for _ in range(k):
                step()
            sync()
            d_ = torch.tensor([(time.perf_counter() - t0_) / k * 1e3], device=dev, dtype=torch.float64)
            dist.all_reduce(d_, op=dist.ReduceOp.MAX)
            return float(d_.item())

        mode_ms[default_name] = dt / args.steps * 1e3
        for name in ('allreduce', 'rsag', 'overlap', 'none'):
            if name != default_name:
                try:
                    mode_ms[name] = timed(name)
                except Exception as e:          # a mode that cannot run here must not cost the line
                    mode_ms[name] = 'failed: %s' % type(e).__name__
        mode_now.update(overlap=overlap, sharded=sharded, none=False)
        return mode_ms

    mode_ms = None
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    captions = B * n * world * args.steps
    value = captions / dt

    if rank == 0:
        copy_gbs = 0.0 if (args.no_prof or args.brief) else measured_copy_gbs(dev)     # kept out of rocprofv3 kernel tables
        # the dominant kernel = ONE instance, gemm_lc_kernel<true,2> (round 3): the decode-step GEMMs that stream >= 24 MB of weights
        # (the language-LSTM gate GEMM, the logit GEMM and the first / un-fused attention-LSTM gate GEMMs; class 9).  The small decode
        # GEMMs (h2att, prepare, and -- r4 -- the 17-MB token-embedding segment left of the attention-LSTM gate GEMM once its h_lang /
        # h_att segments run inside the select launch: class 0) are latency-bound launches of the same template family and are
        # reported together with it under `all_decode_gemms`.
        s_ms, s_n, s_bytes, s_flops = prof_read(lib, 0)
        g_ms, g_n, g_bytes, g_flops = prof_read(lib, 9)
        all_ms, all_n, all_bytes = s_ms + g_ms, s_n + g_n, s_bytes + g_bytes
        a_ms, a_n, a_bytes, _ = prof_read(lib, 3)
        f_ms, f_n, f_bytes, f_flops = prof_read(lib, 2)      # fat GEMMs (sampled only for the XE configurations)
        per_class = {'sampled_steps': graph_sample or sorted(sampled),
                     'gemm_decode_stream': {'ms_per_step': round(g_ms / n_g, 4), 'launches_per_step': g_n / n_g},
                     'gemm_decode_small': {'ms_per_step': round(s_ms / n_extra, 4), 'launches_per_step': s_n / n_extra},
                     'attention_fwd': {'ms_per_step': round(a_ms / n_extra, 4), 'launches_per_step': a_n / n_extra},
                     'companion_sample': 'gemm_decode_small / attention_fwd: one extra step right behind the timed region',
                     'note': 'full per-kernel table: profiles/r06_scst_kernel_stats.md (rocprofv3 --kernel-trace --stats)'}
        ach = (g_bytes / g_n) / (g_ms / g_n * 1e-3) / 1e9 if g_n else 0.0
        tfl = g_flops / (g_ms * 1e-3) / 1e12 if g_ms else 0.0
        # which roof bounds this launch mix.  The decode GEMMs compute fp32 through the bf16 pipe by the exact 3-way
        # split (6 bf16 MFMAs per fp32 MAC tile): fp32-equivalent matrix peak = 2500 / 6 = 417 TFLOP/s, machine balance
        # 52 flop/byte, above the 30 flop/byte of a 60-row weight stream -> HBM-bound.  With CAPMI_ARES_X3=0 they run on
        # the exact-fp32 MFMA (157.3 TFLOP/s, balance 19.7 flop/byte) and are MFMA-bound.
        x3 = os.environ.get('CAPMI_ARES_X3', '1') != '0'
        mfma_peak = MFMA_BF16_PEAK_TFLOPS / 6.0 if x3 else MFMA_F32_PEAK_TFLOPS
        ai = g_flops / g_bytes if g_bytes else 0.0
        mfma_bound = ai > mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        all_ach = (all_bytes / all_n) / (all_ms / all_n * 1e-3) / 1e9 if all_n else 0.0
        lc = os.environ.get('CAPMI_LC', '1') != '0' and os.environ.get('CAPMI_PLANES', '1') != '0' and x3
        roofline = {'kernel': ('gemm_lc_kernel<true,2> (LSTM-gate / logit GEMMs of the decode step: weight streaming, M<=64; loader waves '
                               'copy producer-written bf16x3 activation planes + fp32 weight tiles into a 5-stage LDS ring by LDS-DMA, '
                               'consumer waves split the weights and run v_mfma_f32_32x32x16_bf16)') if lc else
                              ('gemm_ares_kernel<true,6,2> (decode-step GEMMs, activations resident in LDS, %s)'
                               % ('fp32 via exact bf16x3 split, v_mfma_f32_32x32x16_bf16' if x3 else 'v_mfma_f32_32x32x2_f32')),
                    'launches_per_step': g_n / n_g, 'sampled_launches': g_n,
                    'launch_mix_note': 'r4: 41 launches per iteration stream >= 24 MB (this object); 19 attention-LSTM gate GEMMs are down '
                                       'to their 17-MB token-embedding segment because the other 32 MB run inside the select launch '
                                       '(select_gemm_kernel) -- counted in all_decode_gemms; under the round-3 rule (>= 16 MB) frac reads 0.31',
                    'bound': 'mfma' if mfma_bound else 'hbm',
                    'achieved': round(tfl if mfma_bound else ach, 2),
                    'peak': round(mfma_peak, 1) if mfma_bound else HBM_PEAK_GBS,
                    'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
                    'frac': round(tfl / mfma_peak if mfma_bound else ach / HBM_PEAK_GBS, 4),
                    'traffic': pmc_traffic(instance=True),
                    'traffic_source': 'imported: profiles/%s (separate rocprofv3 --pmc passes of this command, scripts/tools_pmc_traffic.py); not measured in this run' % PMC_FILE,
                    'hbm_copy_measured_gbs': round(copy_gbs, 1),
                    'hbm_frac_of_measured_copy': round(ach / copy_gbs, 4) if copy_gbs else None, 'avg_launch_us': round(g_ms / max(g_n, 1) * 1e3, 2),
                    'algorithmic_bytes_per_launch': round(g_bytes / max(g_n, 1)),
                    'algorithmic_flops_per_launch': round(g_flops / max(g_n, 1)),
                    'flop_per_byte': round(ai, 2), 'hbm_gbs': round(ach, 1), 'hbm_frac': round(ach / HBM_PEAK_GBS, 4),
                    'mfma_tflops': round(tfl, 2), 'mfma_frac': round(tfl / mfma_peak, 4), 'mfma_peak_tflops': round(mfma_peak, 1),
                    'all_decode_gemms': {'launches_per_step': s_n / n_extra + g_n / n_g, 'avg_launch_us': round(all_ms / max(all_n, 1) * 1e3, 2),
                                         'algorithmic_bytes_per_launch': round(all_bytes / max(all_n, 1)),
                                         'achieved': round(all_ach, 1), 'frac': round(all_ach / HBM_PEAK_GBS, 4),
                                         'traffic': pmc_traffic(),
                                         'note': 'round-1 definition of this object: the streaming launches above + the '
                                                 '~22 latency-bound small decode GEMMs (h2att, prepare) per step'}}
        a_ach = (a_bytes / a_n) / (a_ms / a_n * 1e-3) / 1e9 if a_n else 0.0
        attention = {'kernel': 'attention_fwd_v2 (fused score + softmax + context; one caption row per workgroup group, the row\'s context columns split over CS workgroups)', 'bound': 'hbm',
                     'achieved': round(a_ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(a_ach / HBM_PEAK_GBS, 4), 'avg_launch_us': round(a_ms / max(a_n, 1) * 1e3, 2),
                     'algorithmic_bytes_per_launch': round(a_bytes / max(a_n, 1)),
                     'note': 'bs10 x n5: 2.4-4.7 MB unique bytes per launch is below the HBM bandwidth-delay product '
                             '(latency-bound); large_batch is the same kernel at an evaluation-sized batch',
                     'large_batch': None if (args.no_prof or args.brief) else attention_large_batch(dev)}
        if attention['large_batch'] and copy_gbs:
            attention['large_batch']['frac_of_measured_copy'] = round(attention['large_batch']['achieved_gbs'] / copy_gbs, 4)
        if args.config in ('updown_xe', 'transformer_xe', 'newfc_xe'):
            # XE steps: the time-batched GEMMs (vocabulary projection over all N*T rows, weight gradients, Transformer QKV / FFN)
            # dominate and are matrix-pipe problems: bf16 MFMA through the exact 3-way split, fp32-equivalent peak 2500 / 6 TF
            peak = MFMA_BF16_PEAK_TFLOPS / 6.0
            tf = f_flops / (f_ms * 1e-3) / 1e12 if f_ms else 0.0
            roofline = {'kernel': 'gemm_x3_kernel / gemm_x3w_kernel (all fat GEMMs of the step, 128 x 128 or 256 x 128 tiles as the planner '
                                  'costs them: fp32 operands split exactly into 3 bf16 planes in the kernel, v_mfma_f32_32x32x16_bf16, '
                                  '6 MFMAs per fp32 MAC tile)',
                        'launches_per_step': f_n / n_sampled, 'sampled_launches': f_n, 'bound': 'mfma', 'achieved': round(tf, 2),
                        'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(tf / peak, 4), 'traffic': None,
                        'avg_launch_us': round(f_ms / max(f_n, 1) * 1e3, 2),
                        'algorithmic_flops_per_launch': round(f_flops / max(f_n, 1)),
                        'algorithmic_bytes_per_launch': round(f_bytes / max(f_n, 1)),
                        'ms_per_step': round(f_ms / n_sampled, 4),
                        'note': 'fp32-equivalent FLOPs (2 M N K) over the in-dispatch HIP-event time of every fat GEMM launch of the '
                                'sampled steps; the bf16 pipe executes 6x as many'}
            if args.config == 'transformer_xe' and os.environ.get('CAPMI_DW_STREAM', '0') == '1':
                roofline['concurrency_note'] = ('r4: the weight-gradient GEMMs run on a side stream beside the dX chain (ops.DeferredGrads), so fat '
                                                'GEMMs overlap and each launch takes longer than it does alone: the step is 7 % faster, the per-launch '
                                                'rate reads lower (0.28 with CAPMI_DW_STREAM=0, profiles/r04e_txe_kernel_stats.md)')
        if args.config not in ('updown_scst', 'updown_xe'):
            attention = None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = (cpu_baseline(opt, model, B, n, L, 1 if args.brief else args.cpu_iters) if args.config == 'updown_scst'
                   else cpu_baseline_other(args.config, opt, model, B, L, brief=args.brief))
        early = None
        if args.config == 'updown_scst' and world == 1 and not args.no_prof and not args.brief and os.environ.get('CAPMI_EARLY_EXIT', '4') != '0':
            early = early_exit_line(model, flat, lw, pf, gt_indices, opt, B, n)
        names = {'updown_scst': 'captions/sec/node (UpDown SCST, bs10xsample_n5, 36x2048 feats)',
                 'updown_xe': 'captions/sec/node (UpDown XE, bs64 x 5 captions, 36x2048 feats)',
                 'transformer_xe': 'captions/sec/node (Transformer XE, bs64 x 5 captions, 36x2048 feats)',
                 'aoa_nsc': 'captions/sec/node (AoA new-self-critical, bs10xsample_n5, 36x2048 feats)',
                 'newfc_xe': 'captions/sec/node (NewFC XE, bs10 x 5 captions, 2048-d fc feats)'}
        line = {
            'metric': names[args.config], 'value': round(value, 2),
            'unit': 'captions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'init_steps': INIT_STEPS,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'step_ms': step_ms, 'higher_is_better': True,
            'scaling': 'strong' if args.global_batch else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'numerics': 'fp32 storage and accumulation; GEMMs on the bf16 matrix pipe through an exact 3-way operand split '
                        '(fp32-grade error, DESIGN.md 4); CAPMI_GEMM_X3=0 CAPMI_ARES_X3=0 selects the exact-fp32 MFMA',
            'config': {'workload': cfg[4], 'baseline_config': 'BASELINE.json configs[%d]' % cfg[3],
                       'rotating_batches': NB, 'global_batch': B * world, 'captions_per_step': B * n * world, 'seq_len': L,
                       'parallelism': 'dp%d (flat fp32 gradient, %s)' % (world, ('bucketed RCCL all-reduce overlapped with the backward' if overlap else 'one RCCL all-reduce per step') if multi else 'no collective')},
            'collective': None if not multi else {'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                                                   'mode': 'bucketed overlap' if overlap else ('reduce-scatter + sharded Adam + all-gather' if sharded else 'one flat all-reduce per step'),
                                                   'bytes': int(flat.grad.numel() * 4), 'allreduce_ms': None if allreduce_ms is None else round(allreduce_ms, 3),
                                                   'ms_per_step_by_mode': None if mode_ms is None else {k: (round(v, 3) if isinstance(v, float) else v) for k, v in mode_ms.items()},
                                                   'exposed_comm_ms': None if not (mode_ms and isinstance(mode_ms.get('none'), float)) else
                                                   {k: round(v - mode_ms['none'], 3) for k, v in mode_ms.items() if isinstance(v, float) and k != 'none'},
                                                   'note': 'ms_per_step_by_mode: the default mode over the K timed steps, the others over 6 extra steps each; '
                                                           '"none" = no gradient exchange (compute only); exposed = mode - none'},
            'step_issue': {'mode': 'hipGraph replay (graph_step.TrainStep)' if ts.replays else 'stepped launch by launch',
                           'captures': ts.captures, 'replays': ts.replays, 'stepped': ts.stepped, 'capture_failed': ts.failed},
            'loss': float(loss.detach()), 'roofline': roofline, 'attention': attention, 'kernel_ms_per_step': per_class,
            'early_exit_eos_biased': early, 'cpu_baseline': cpu}
        WATCH['line'] = line
        if (args.config == 'updown_scst' and world == 1 and not (args.brief or args.no_prof or args.no_other_configs)
                and os.environ.get('CAPMI_BENCH_OTHER_CONFIGS', '1') != '0'):
            # every BASELINE configuration on the driver's line, measured after the headline is complete
            line['other_configs'] = other_configs()
    if multi and os.environ.get('CAPMI_BENCH_MODES', '1') != '0':
        # every exchange mode in the same run, AFTER the measurement and its line are complete: if a mode hangs on a topology that
        # could not be tested here, the watchdog prints the line without the sweep instead of losing it
        WATCH['phase'] = 'mode sweep'
        mode_ms = run_mode_sweep()
        WATCH['phase'] = 'done'
        if rank == 0:
            c = line['collective']
            c['ms_per_step_by_mode'] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in mode_ms.items()}
            if isinstance(mode_ms.get('none'), float):
                c['exposed_comm_ms'] = {k: round(v - mode_ms['none'], 3) for k, v in mode_ms.items() if isinstance(v, float) and k != 'none'}
    if multi and args.config == 'updown_scst' and not (args.brief or args.no_other_configs) \
            and os.environ.get('CAPMI_BENCH_DDP_CONFIGS', '1') != '0':
        # ONE N-GPU run answers everything -- the two configurations BASELINE.json names for 8-GPU data parallel,
        # on this process group, after the headline line and the sweep are complete (a hang here: the watchdog prints the line as it is)
        WATCH['phase'] = 'mode sweep'           # (same watchdog semantics: the measurement is done)
        WATCH['mode'] = 'ddp other configs'
        del lw, ts
        oc = ddp_other_configs(dev, world, rank, local, dist, df_ref=df_ref)
        WATCH['phase'] = 'done'
        if rank == 0:
            line['other_configs'] = oc
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(opt, model, B, n, L, iters):
    """The oracle's SCST iteration (a port of the reference's CPU path) on this box's host cores: a
    bounded sample of the SAME workload (bs10 x n5, L=20), 1 warm-up + `iters` timed iterations."""
    from oracle import scst_step, ciderd as OC
    from imagecaptioning.pytorch_amd import synthetic
    # The reference's CPU path is torch fp32 on small (M <= 60) matrices: it stops scaling beyond ~16 threads
    # (measured on the 256-core GPU box: 16 thr 1.25 s/iter, 32 thr 2.0 s, 64 thr 3.3 s, 256 thr > 60 s), so the
    # baseline uses the best setting, min(cores, 16), and reports exactly that thread count.
    cores = min(os.cpu_count() or 1, int(os.environ.get('CAPMI_CPU_THREADS', '16')))
    torch.set_num_threads(cores)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    corpus = synthetic.corpus(DF_IMAGES, seed=7)
    df, ref_len = synthetic.document_frequency(corpus)
    oracle = scst_step.ScstOracle(P, OC.CiderD(df, ref_len), drop_prob=opt.drop_prob_lm, lr=opt.learning_rate,
                                  clip=opt.grad_clip_value, sample_n=n, max_len=L)
    fc, att = synthetic.batch(B, seed=1234)
    gts = synthetic.corpus(B, seed=100)
    sec = scst_step.time_iterations(oracle, fc, att, gts, iters=iters, warmup=1)
    out = {'value': round(B * n / sec, 2), 'unit': 'captions/s', 'cores': cores, 'kind': 'port',
           'sample': '%d timed SCST iterations (bs%d x n%d, L=%d) of oracle/scst_step.py after 1 warm-up, median; '
                     'torch fp32 on %d threads (host has %d cores; more threads are slower for these shapes)'
                     % (iters, B, n, L, cores, os.cpu_count() or 1), 'sec_per_iteration': round(sec, 3)}
    # /root/reference does not exist on the GPU box, so the IMPORTED reference modules were timed in the build container on the
    # same workload (scripts/time_reference_cpu.py -> profiles/r02_cpu_reference.json); reported beside the on-box port
    try:
        with open(os.path.join(ROOT, 'profiles', 'r02_cpu_reference.json')) as f:
            out['reference_in_build_container'] = json.load(f)
    except (OSError, ValueError):
        out['reference_in_build_container'] = None
    return out


def early_exit_line(model, flat, lw, pf, gt_indices, opt, B, n, steps=20):
    """Second, separately labelled measurement of the SAME step with a model that ends its captions: random-init weights never
    draw the EOS, so the headline loop always runs all 20 steps and the early exit of the rollout driver (AttModel.py:349-350)
    cannot show.  +12 on logit.bias[EOS] makes every row stop within a few steps; the weights are restored afterwards."""
    bias0 = model.logit.bias.detach()[0].clone()
    with torch.no_grad():
        model.logit.bias[0] += 12.0

    def step():
        data = pf.get_batch('train')
        out = lw(data['fc_feats'], data['att_feats'], None, None, None, data['gts'], gt_indices, True, False, False)
        flat.zero_grad()
        out['loss'].backward()
        flat.collect_grads()
        flat.adam_step(lr=0.0, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon, weight_decay=0.0,
                       clip_value=opt.grad_clip_value, grad_scale=1.0)

    for _ in range(8):             # (the shorter rollouts have their own buffer shapes: let the caching allocator settle)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ro = getattr(model, '_last_rollout', None)
    with torch.no_grad():
        model.logit.bias[0] = bias0
    return {'what': 'the same SCST iteration with +12 on logit.bias[EOS] (every caption ends within a few tokens): the rollout driver '
                    'stops enqueuing decode steps once no row is left and the BPTT runs over the steps that were enqueued',
            'ms_per_step': round(dt * 1e3, 3), 'captions_per_s': round(B * n / dt, 1),
            'decode_steps_enqueued': None if ro is None else int(ro.steps_run), 'decode_steps_max': int(opt.max_length),
            'check_every': int(os.environ.get('CAPMI_EARLY_EXIT', '4'))}


def cpu_baseline_other(config, opt, model, B, L, brief=False):
    """CPU baseline of the non-headline configurations: the matching oracle (oracle/att_lstm.py, transformer.py, aoa.py: ports of
    the reference's CPU path) running the SAME workload at the SAME batch as the GPU step beside it --
    XE configurations: teacher-forced forward + criterion + autograd backward + value clip + Adam on B images x 5 captions;
    aoa_nsc: the actual new-self-critical iteration (loss_wrapper.py:25-48, losses.py:168-187): a SAMPLED rollout of
    train_sample_n rows per image with its autograd graph, CIDEr-D scores on the host, the structure loss, backward, clip, Adam.
    ONE timed iteration after a warm-up on 2 images (threads, allocator): a bs64 iteration takes 10-30 s of host time."""
    from oracle import att_lstm as O, transformer as T, aoa as A, ciderd as OC
    from imagecaptioning.pytorch_amd import synthetic
    cores = min(os.cpu_count() or 1, int(os.environ.get('CAPMI_CPU_THREADS', '16')))
    torch.set_num_threads(cores)
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    optim = torch.optim.Adam(list(P.values()), lr=opt.learning_rate)
    n = 5
    scorer = None
    if config == 'aoa_nsc':
        df, ref_len = synthetic.document_frequency(synthetic.corpus(DF_IMAGES, seed=7))
        scorer = OC.CiderD(df, ref_len)

    def it(Bc):
        fc, att = synthetic.batch(Bc, seed=1234)
        if config == 'aoa_nsc':
            gts = synthetic.corpus(Bc, seed=100)
            seq, logp = A.sample(P, att, None, 8, L, n=n)
            scores = OC.sample_scores(scorer, gts, seq.numpy())                  # rewards.py:83-114 get_scores, on the host
            loss = O.new_self_critical_loss(logp, seq, torch.as_tensor(scores).double().reshape(-1), n)
        else:
            labels, masks = synthetic.xe_labels(Bc, n=5, L=L)
            if config == 'updown_xe':
                logp = O.forward_teacher(P, fc, att, labels[..., :-1], None)
            elif config == 'newfc_xe':
                logp = O.newfc_forward_teacher(P, fc, labels[..., :-1])
            else:
                logp = T.forward_teacher(P, att, labels[..., :-1], None, h=8, n_enc=6, n_dec=6)
            loss = O.lm_criterion(logp, labels[..., 1:], masks[..., 1:])
        optim.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(list(P.values()), opt.grad_clip_value)
        optim.step()

    it(2)
    t0 = time.perf_counter()
    it(B)
    sec = time.perf_counter() - t0
    what = ('a sampled rollout of 5 rows per image with its autograd graph + CIDEr-D scores on the host + new_self_critical loss'
            if config == 'aoa_nsc' else 'teacher-forced forward + criterion')
    return {'value': round(B * 5 / sec, 2), 'unit': 'captions/s', 'cores': cores, 'kind': 'port',
            'sample': 'ONE timed iteration of the %s oracle at the GPU step\'s own batch (%d images x 5 captions) after a 2-image warm-up: '
                      '%s + autograd backward + clip + Adam; torch fp32 on %d threads' % (config, B, what, cores),
            'sec_per_iteration': round(sec, 3)}


if __name__ == '__main__':
    main()
