#!/usr/bin/env python
"""bench.py -- frame-pairs/s of the --mode flow train step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: forward (3B pyramid, 2B decoder, HIP corr /
warp / losses), weighted loss, backward, gradient all-reduce (N > 1, RCCL) and Adam, on
synthetic KITTI-sized triplets [8,3,768,832] fp32 already resident in HBM (BASELINE config 2:
832x256, bs=8 per GPU, fp32; weak scaling: the per-GPU batch is fixed).  pairs/s = 2 * triplets/s.

`python bench.py --gpus N` starts its N ranks itself when no launcher did (unopticalflow_amd/launch.py; the parent makes no GPU
call).  The timed steps REPLAY the step as a hipGraph (`step_mode`; ~3000 launches cost 14-24 ms of host time per step
depending on the box against 24.3 ms of GPU work: an eager loop is host-paced on a slow host, and always in bf16) -- with several
ranks as graph(forward + backward + gradient pack) -> one RCCL all-reduce -> graph(Adam); `--graph 0` runs the steps eagerly
(several ranks: gradient pieces leave for RCCL from hooks during backward).

The single JSON line also carries
  roofline:     HIP events around EVERY cost-volume / warp entry point inside the timed steps (on the launch stream); under replay
                (a captured graph cannot carry per-launch events) over the same number of eager steps right behind them
                (`measured_in`).
                The object describes the (entry point, shape) with the LARGEST time per step -- the kernel that weighs most,
                not the one that looks best: algorithmic bytes per launch (SURVEY 8d formulas) / mean launch duration
                against the 8 TB/s HBM peak.  `traffic` = HBM bytes per launch from the PMC passes committed under
                profiles/ -- only when that file was measured on the very kernel sources of this build (sha256 of
                csrc/*.hip recorded next to it), else null.  `aggregate` = all cost-volume + warp launches of a step as
                one figure (north_star: "achieved HBM GB/s for corr/warp") with the per-level table; `best` = the entry
                with the highest fraction, for reference; `losses` = the same aggregate + per-entry table for the loss kernels
                (SSIM window reduction, occlusion weights, masked L1, smoothness, consistency);
  cpu_baseline: the CPU oracle (oracle/ref_cpu.py, the restatement of the reference's op graph,
                kind "port") timed on the host cores of this box on a bounded sample (all cores the
                cgroup allows, plus a one-thread figure);
  conv_stack:   convolution FLOPs of the step / whole step time / dense MFMA peak of the dtype: a lower
                bound on the conv stacks' MFMA utilisation; `conv_kernels_only`: the convolution kernels' own utilisation from
                the MFMA hardware counters of a committed rocprofv3 --pmc capture (profiles/r3_conv_mfma_*.json);
  step_ms, host_enqueue_ms, drain_ms, all_step_ms, host_gc: the spread of the timed steps -- device time between per-step
                event records, host time to enqueue a step, how far the host was ahead at the end, and what Python's
                cyclic collector did inside the timed region (one generation-2 pass of ~90 ms used to put a 50-90 ms step
                into every ~17: the trainer now freezes the heap after its second step, FlowTrainer(gc_freeze_after));
  kernel_survey (N = 1 only, 3 extra untimed steps with HIP events around every C entry point): the 14 heaviest
                entry points (losses and conv epilogues included).
--force-ddp runs the N = 1 step through the data-parallel path on RCCL with a one-rank communicator (hooks, async
all-reduce on RCCL's stream, waits): a rehearsal of what every rank of an 8-GPU job executes.
UNFLOW_BENCH_ONE_GPU=1 is a rehearsal mode for a single-GPU box (all ranks on device 0, gloo).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def self_launch_if_needed():
    """``python bench.py --gpus N`` (N > 1) with no launcher in front: start the N ranks ourselves (the reference takes its GPU
    list on one command line, train.py:208-214).  Runs BEFORE torch is imported: the parent never initialises HIP, it relays
    rank 0's JSON line as its own last stdout line and exits with the worst child code.  UNFLOW_BENCH_ONE_GPU=1 (rehearsal on a
    one-GPU box) goes the same way."""
    n, argv = 1, sys.argv[1:]
    for i, a in enumerate(argv):
        if a == '--gpus' and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith('--gpus='):
            n = int(a.split('=', 1)[1])
    if n <= 1 or 'WORLD_SIZE' in os.environ:
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location('_unflow_launch', os.path.join(ROOT, 'unopticalflow_amd', 'launch.py'))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)                              # (by path: importing the package would import torch)
    sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n))


if __name__ == '__main__':
    self_launch_if_needed()

import torch                                                     # noqa: E402
import torch.distributed as dist                                 # noqa: E402

H, W, B_PER_GPU = 256, 832, 8
MFMA_PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0}   # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=B_PER_GPU, help='triplets per GPU (8 = BASELINE config)')
    ap.add_argument('--hw', type=int, nargs=2, default=[H, W], metavar=('H', 'W'),
                    help='frame size (default 256 832 = the BASELINE metric; 448 1024 with --batch 4 = config 4; both are in the shipped find-db)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=8, help='triplets in the CPU-baseline sample step (8 = the bench batch)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'], help='conv-stack precision (bf16: BASELINE config 3; the headline metric is fp32)')
    ap.add_argument('--graph', type=int, default=-1, help='1: replay the step as a hipGraph; 0: eager (per-launch events INSIDE the timed steps); default: replay.  Under replay the roofline legs are taken over the same number of EAGER steps (at most 20) right behind the timed region: a captured graph cannot carry per-launch event pairs')
    ap.add_argument('--channels-last', type=int, default=-1, help='memory format of the conv stacks: 1 channels_last (NHWC), 0 NCHW; default: 1 when the shipped MIOpen find-db is in use, else 0 (cfg.channels_last)')
    ap.add_argument('--weight-shadows', type=int, default=1, help='bf16 only; 0: autocast casts every convolution weight per call instead of one multi-tensor cast per pass (A/B; cfg.weight_shadows)')
    ap.add_argument('--fused-levels', default='', help="decoder levels (e.g. '5' or '4,5') whose warp + cost volume run as the one fused kernel (cfg.fused_warp_corr_levels); --fused 1 = all four")
    ap.add_argument('--fused', type=int, default=0, help='1: warp + cost volume of each decoder level as one kernel (cfg.fused_warp_corr)')
    ap.add_argument('--dup-centre', type=int, default=1, help='0: torch.cat((c, c)) of the centre features instead of the hand-off that writes them twice (A/B; Model_flow.dup_centre)')
    ap.add_argument('--fused-head', type=int, default=1, help='0: ATen bias add, re-layout copy and residual add behind the flow heads instead of unflow_flow_head_* (A/B; PWC_tf.fused_head)')
    ap.add_argument('--fused-loss-sums', type=int, default=1, help='0: eager adds / means for the loss bookkeeping instead of unflow_loss_combine_* and unflow_weighted_mean_sum_* (A/B)')
    ap.add_argument('--fused-upsample', type=int, default=1, help='0: F.interpolate + multiply for the flow up-sampling instead of unflow_upsample_scaled_* (A/B; PWC_tf.fused_upsample)')
    ap.add_argument('--fill-cat', type=int, default=1, help='0: channels_last decoder with torch.cat inputs instead of epilogue-filled cat buffers (A/B; PWC_tf.fill_cat_buffers)')
    ap.add_argument('--fused-warp-bwd', type=int, default=1, help='0: zero-fill + scatter for the feature-map warps\' backward instead of the one-pass gather (A/B; ops.fused_warp_bwd)')
    ap.add_argument('--split-handoff', type=int, default=0, help='1: the pyramid hand-off returns both decoder inputs itself (Model_flow.split_handoff: no split, no gradient concatenation; A/B)')
    ap.add_argument('--multiscale-losses', type=int, default=0, help='1: every loss of the scale loop as one launch over the three scales (A/B; Model_flow.multiscale_losses, off until GPU-validated)')
    ap.add_argument('--deferred-loss-sums', type=int, default=0, help='1: ONE second-stage launch per forward pass for all per-sample loss reductions instead of one per reduction (A/B; Model_flow.deferred_loss_sums, off until a complete GPU suite has covered it)')
    ap.add_argument('--corr-bwd', default='auto', choices=['auto', 'fp32', 'mfma', 'fp32_next'], help='cost-volume backward arithmetic, per call (PWC_tf.corr_backward -> unflow_corr_bwd_ex): mfma = the matrix-core form at d = 4 too')
    ap.add_argument('--gc-freeze', type=int, default=1, help='0: leave Python\'s cyclic collector alone; 1: FlowTrainer(gc_freeze_after=2), what train.py asks for too (gc.freeze() after the second step, once per process)')
    ap.add_argument('--contended-host', action='store_true', help='experiment (profiles/r4_multirank_step_mode.md): for the TIMED steps confine this process to one core and run a busy-loop child on the same core -- what a slow or shared host does to the step mode')
    ap.add_argument('--force-ddp', action='store_true', help='N = 1 only: run the step through the RCCL data-parallel path with a one-rank communicator')
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sample_b):
    """The oracle's train step (same op graph as the reference: 81-offset corr loop, grid_sample,
    AvgPool SSIM) on this box's host cores: 1 untimed + 3 timed steps of `sample_b` triplets."""
    from oracle import ref_cpu as R
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = R.default_cfg()
    model = R.Model_flow(cfg)
    model.load_state_dict(R.seeded_state_dict(model, 1234, 0.25))
    opt = torch.optim.Adam([{'params': [p for p in model.parameters() if p.requires_grad], 'lr': cfg.lr}])
    weights = R.generate_loss_weights_dict(cfg)
    x = R.synthetic_triplets(sample_b, H, W, seed=0, structured=False)
    R.train_step(model, opt, x, weights)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        R.train_step(model, opt, x, weights)
    dt = time.perf_counter() - t0
    # single-thread figure: one timed step of one triplet after one untimed step
    torch.set_num_threads(1)
    x1 = x[:1].contiguous()
    R.train_step(model, opt, x1, weights)
    t1 = time.perf_counter()
    R.train_step(model, opt, x1, weights)
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(cores)
    cpu = 'unknown CPU'
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                cpu = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': round(2 * sample_b * n / dt, 4), 'unit': 'pairs/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'value_1thread': round(2 / dt1, 4),
            'sample': '%d timed train steps (+1 untimed) of %d synthetic 832x256 triplets (%.1f s), fp32, torch CPU '
                      'oracle on %s; value_1thread: 1 timed step of 1 triplet on one thread (%.1f s)' % (n, sample_b, dt, cpu, dt1)}


def per_step_stats(marks, host, t_end):
    """Spread of the timed steps.  `step_ms` = device time between consecutive per-step event records (what the GPU
    spent on a step, including any wait for the host), `host_enqueue_ms` = host time to ENQUEUE a step (no sync inside),
    `drain_ms` = time between the last enqueue and the closing synchronisation: when the drain is far below one step the
    host is what paces the loop (launch-bound), when it is several steps the GPU is."""
    import statistics as st
    dev = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
    enq = [(host[i + 1] - host[i]) * 1e3 for i in range(len(host) - 1)]
    k = min(5, len(dev))

    def summary(v):
        s = sorted(v)
        return {'min': round(s[0], 3), 'median': round(st.median(s), 3), 'p90': round(s[min(len(s) - 1, int(0.9 * len(s)))], 3),
                'max': round(s[-1], 3), 'first5_mean': round(sum(v[:k]) / k, 3), 'last5_mean': round(sum(v[-k:]) / k, 3)}
    return {'step_ms': summary(dev), 'host_enqueue_ms': summary(enq), 'drain_ms': round((t_end - host[-1]) * 1e3, 3),
            'all_step_ms': [round(x, 2) for x in dev]}


def sources_sha16():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from, in name order."""
    import hashlib
    d = os.path.join(ROOT, 'unopticalflow_amd', 'csrc')
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h', '.cpp')):
            h.update(name.encode()); h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


def host_sources_sha16():
    """sha256 (first 16 hex digits) over the Python sources that decide WHICH convolutions a step runs (the package, not tools)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'unopticalflow_amd')
    for d, _, files in sorted(os.walk(base)):
        for name in sorted(files):
            if name.endswith('.py'):
                h.update(os.path.relpath(os.path.join(d, name), base).encode()); h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


def latest_profile(pattern):
    """profiles/r<N>_<pattern> of the highest round present (this round's capture when there is one, else the last one)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_' + pattern)):
        m = re.match(r'r(\d+)_', os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def measured_traffic(entry, shape, path=None):
    """HBM bytes per launch of (entry point, shape) from the rocprofv3 PMC passes under profiles/ (FETCH_SIZE x2 per the
    gfx950 wide-load correction + WRITE_SIZE, tools/pmc_traffic.py) -- but only if that file was measured on exactly the
    kernel sources of this build; a stale file gives (None, reason)."""
    path = path or latest_profile('pmc_traffic.json')
    if path is None or not os.path.exists(path):
        return None, 'no PMC file'
    d = json.load(open(path))
    key = '%s %s' % (entry, list(shape) if shape else None)
    v = d.get('entries', {}).get(key)
    if d.get('sources_sha16') != sources_sha16():
        # other sources -- but the same MACHINE CODE?  The file names the kernels that served the entry and (kernel_isa) the hash of each
        # one's instruction stream in the measured build; if hipcc emits byte-identical kernels from today's sources the measurement stands
        # (a comment or a pruned template parameter changes the source hash, not the traffic)
        same, why = _kernels_unchanged(d, v)
        if not same:
            return None, 'PMC file %s was measured on other kernel sources (%s != %s; %s)' % (os.path.basename(path), d.get('sources_sha16'), sources_sha16(), why)
        return int(v['hbm_bytes_per_launch']), 'profiles/%s (measured on sources %s; the kernels serving this entry are byte-identical in this build: %s)' % (
            os.path.basename(path), d['sources_sha16'], why)
    if v is None:
        return None, 'no PMC entry for %s' % key
    return int(v['hbm_bytes_per_launch']), 'profiles/%s (sources %s)' % (os.path.basename(path), d['sources_sha16'])


def _kernels_unchanged(pmc, entry_row):
    """(True, 'kernel sha16, ...') when every kernel that served `entry_row` in the PMC capture has, compiled from today's sources with
    the build's flags and the same hipcc, the instruction-stream hash the capture's sources gave (tools/isa_hashes.py)."""
    isa = (pmc.get('kernel_isa') or {})
    if entry_row is None or not isa.get('kernels'):
        return False, 'no per-kernel hashes in the file'
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import isa_hashes
        if isa_hashes.hipcc_version() != isa.get('hipcc') or ' '.join(isa_hashes.build_flags()) != isa.get('flags'):
            return False, 'another compiler or other flags'
        seen = []
        for k in entry_row.get('kernels', {}):
            ref = isa['kernels'].get(k)
            if ref is None:
                return False, 'no hash for %s' % k
            now = isa_hashes.isa_hashes(os.path.join(ROOT, 'unopticalflow_amd', 'csrc', ref['file']))
            names = isa_hashes.demangle(list(now))
            got = {isa_hashes.short_name(names[m]): v['sha16'] for m, v in now.items()}.get(k)
            if got != ref['sha16']:
                return False, '%s differs (%s != %s)' % (k, got, ref['sha16'])
            seen.append('%s %s' % (k, got))
        return bool(seen), ', '.join(seen)
    except Exception as e:                                     # noqa: BLE001  (no hipcc on the box, ...: the traffic stays null)
        return False, 'ISA comparison unavailable (%s: %s)' % (type(e).__name__, e)
    finally:
        sys.path.pop(0)


def main():
    args = parse()
    auto_graph = args.graph < 0
    if args.graph < 0:
        # replay, with one process and with several.  The step's ~3000 launches take 14-24 ms of host time depending on the box,
        # against 24 ms (fp32) / 11 ms (bf16) of GPU work: the eager bf16 step is host-bound everywhere, the fp32 step on a slow or
        # contended host -- and with N ranks the slowest host paces all of them.  Replayed, a rank enqueues an input copy, two graph
        # launches and ONE 20.5 MB all-reduce per step (FlowTrainer._build_graph); the exchange is not overlapped with backward
        # (~0.3 ms exposed of a 24 ms step) but no rank ever waits for a Python interpreter.  --graph 0: eager launches, gradient
        # pieces all-reduced from hooks during backward.  A/B incl. a contended host: profiles/r4_multirank_step_mode.md
        args.graph = 1
    from unopticalflow_amd import get_model, _lib, ops
    from unopticalflow_amd.parallel import init_distributed
    from unopticalflow_amd.trainer import FlowTrainer
    import types

    _lib.load()                                   # no HIP library -> fail loudly, never fall back
    ops.fused_warp_bwd = bool(args.fused_warp_bwd)
    # one rank per GPU over RCCL.  UNFLOW_BENCH_ONE_GPU=1 is a rehearsal mode for boxes with a single GPU: every rank
    # shares device 0 and the collectives go through gloo (RCCL refuses two ranks on one device) -- it exercises the
    # multi-rank plumbing, its numbers mean nothing.
    one_gpu = os.environ.get('UNFLOW_BENCH_ONE_GPU') == '1'
    rank, local_rank, world = init_distributed('gloo' if one_gpu else 'nccl', device_index=0 if one_gpu else None,
                                               force=args.force_ddp)
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from unopticalflow_amd import tuning
    if os.environ.get('UNFLOW_MIOPEN_FIND', '1') == '1':
        tuning.enable_miopen_tuning()             # shipped find-db for exactly these conv shapes (tuning.py)
    measured_picks = tuning.default_channels_last()
    # channels_last conv stacks pay off with MIOpen's MEASURED solver picks (the shipped find-db: fp32 26.1 -> 25.3 ms,
    # bf16 15.9 -> 14.1 ms); with
    # its immediate-mode heuristics (another MIOpen build than the db was made with) NHWC is the slower layout (29.2 vs
    # 27.3 ms), so the default follows whether the db is in use
    cl = measured_picks if args.channels_last < 0 else bool(args.channels_last)
    cfg = types.SimpleNamespace(mode='flow', dataset='kitti_depth', num_scales=3, h_flow_consist_alpha=3.0,
                                h_flow_consist_beta=0.05, w_ssim=0.85, w_flow_smooth=10.0, w_flow_consis=0.01,
                                lr=1e-4, align_corners=False, precision=args.precision, weight_shadows=bool(args.weight_shadows), fused_warp_corr=bool(args.fused), fused_warp_corr_levels=(args.fused_levels or None),
                                channels_last=cl)
    torch.manual_seed(1234)                       # same random init on every rank
    model = get_model('flow')(cfg).to(dev)
    model.pwc_model.fill_cat_buffers = bool(args.fill_cat)
    model.pwc_model.fused_upsample = bool(args.fused_upsample)
    model.pwc_model.fused_head = bool(args.fused_head)
    model.fused_loss_sums = bool(args.fused_loss_sums)
    model.deferred_loss_sums = bool(args.deferred_loss_sums)
    model.multiscale_losses = bool(args.multiscale_losses)
    model.pwc_model.corr_backward = args.corr_bwd
    model.split_handoff = bool(args.split_handoff)
    model.dup_centre = bool(args.dup_centre)
    trainer = FlowTrainer(cfg, model, distributed=(world > 1 or args.force_ddp), use_graph=bool(args.graph),
                          single_rank_collectives=args.force_ddp, gc_freeze_after=2 if args.gc_freeze else None)
    trainer.fused_total_loss = bool(args.fused_loss_sums)
    user_no_timing = args.no_kernel_timing
    graph_timing = bool(args.graph) and not args.no_kernel_timing      # replayed timed region: the per-launch events need eager steps (after it)
    if args.graph:
        args.no_kernel_timing = True
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)                         # distinct synthetic data per rank
    fh, fw = args.hw
    inputs = torch.rand((args.batch, 3, 3 * fh, fw), generator=gen, device=dev, dtype=torch.float32)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    warm = args.warmup
    if args.graph and auto_graph and warm > 0:
        # the DEFAULT replay mode must not cost a run: if the capture fails on this box (it builds inside the first step), say so
        # and fall back to the eager step on a fresh model / trainer -- an explicit --graph 1 fails loudly instead
        try:
            trainer.step(inputs)
            torch.cuda.synchronize()
            warm -= 1
        except Exception as e:                              # noqa: BLE001
            sys.stderr.write('bench.py: hipGraph capture failed (%s: %s); continuing with eager steps\n' % (type(e).__name__, e))
            args.graph, graph_timing, args.no_kernel_timing = 0, False, user_no_timing
            torch.manual_seed(1234)
            model = get_model('flow')(cfg).to(dev)
            model.pwc_model.fill_cat_buffers = bool(args.fill_cat)
            model.pwc_model.fused_upsample = bool(args.fused_upsample)
            model.pwc_model.fused_head = bool(args.fused_head)
            model.fused_loss_sums = bool(args.fused_loss_sums)
            model.deferred_loss_sums = bool(args.deferred_loss_sums)
            model.multiscale_losses = bool(args.multiscale_losses)
            model.pwc_model.corr_backward = args.corr_bwd
            model.split_handoff = bool(args.split_handoff)
            model.dup_centre = bool(args.dup_centre)
            trainer = FlowTrainer(cfg, model, distributed=(world > 1 or args.force_ddp), use_graph=False,
                                  single_rank_collectives=args.force_ddp, gc_freeze_after=2 if args.gc_freeze else None)
            trainer.fused_total_loss = bool(args.fused_loss_sums)
    for _ in range(warm):
        trainer.step(inputs)
    CW = ('unflow_corr_fwd', 'unflow_corr_bwd', 'unflow_warp_fwd', 'unflow_warp_fwd_table', 'unflow_warp_bwd', 'unflow_warp_bwd_det', 'unflow_warp_bwd_fused',
          'unflow_warp_corr_fwd', 'unflow_warp_corr_bwd', 'unflow_warp_fwd_ms', 'unflow_warp_bwd_ms')
    # the second timed set: the occlusion-aware loss kernels (north_star names the SSIM window reduction and the occlusion-mask
    # ops next to corr / warp) -> roofline.losses
    LOSSES = ('unflow_ssim_loss_fwd', 'unflow_ssim_loss_bwd', 'unflow_occ_weight_fwd', 'unflow_absdiff_bwd', 'unflow_masked_mean_fwd',
              'unflow_masked_mean_bwd', 'unflow_smooth2_fwd', 'unflow_smooth2_bwd', 'unflow_consis_fwd', 'unflow_consis_bwd')
    LOSSES += tuple(e + '_ms' for e in LOSSES)           # (--multiscale-losses 1: the same ten entries, one launch over the scales each)
    TIMED = CW + LOSSES
    if not args.no_kernel_timing:
        ops.kernel_timer.enable(TIMED, reserve=96 * args.steps)     # every such launch of the timed steps: kernel-exact event pairs
    # per-step clocks that do not perturb the loop: one event record per step on the launch stream (read after the
    # closing barrier) and one host stamp per step (no synchronisation inside the timed region)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host = [0.0] * (args.steps + 1)
    import gc
    gc_log = {'collections': [0, 0, 0], 'ms': 0.0, 'max_ms': 0.0, '_t': 0.0}

    def gc_watch(phase, info):                        # how much of the timed region the cyclic collector took (host side)
        if phase == 'start':
            gc_log['_t'] = time.perf_counter()
        else:
            d = (time.perf_counter() - gc_log['_t']) * 1e3
            gc_log['collections'][info['generation']] += 1
            gc_log['ms'] += d
            gc_log['max_ms'] = max(gc_log['max_ms'], d)
    gc.callbacks.append(gc_watch)
    busy, affinity = None, None
    try:
        if args.contended_host:
            import subprocess
            affinity = os.sched_getaffinity(0)                # the mask this process was STARTED with (cpuset / taskset): restored below
            core = sorted(affinity)[0]
            os.sched_setaffinity(0, {core})
            busy = subprocess.Popen([sys.executable, '-c', 'import os\nos.sched_setaffinity(0, {%d})\nwhile True: pass' % core])
            time.sleep(0.5)
        barrier()
        t0 = time.perf_counter()
        marks[0].record()
        host[0] = t0
        for i in range(args.steps):
            loss, _ = trainer.step(inputs)
            marks[i + 1].record()
            host[i + 1] = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
    finally:                                                  # also when a step raises: no busy loop left behind, the rest of the run on the old mask
        if busy is not None:
            busy.kill(); busy.wait()                          # (our own child, by handle)
        if affinity is not None:
            os.sched_setaffinity(0, affinity)
    gc.callbacks.remove(gc_watch)
    step_stats = per_step_stats(marks, host, t0 + dt)
    step_stats['host_gc'] = {'collections_gen0_1_2': gc_log['collections'], 'total_ms': round(gc_log['ms'], 2),
                             'longest_ms': round(gc_log['max_ms'], 2), 'frozen_by_trainer': bool(trainer._gc_frozen)}
    ops.kernel_timer.disable()
    timed_rows = ops.kernel_timer.rows() if not args.no_kernel_timing else []      # (device is synchronised: barrier())
    roofline_steps = float(args.steps)
    if graph_timing:
        # a captured graph cannot carry the per-launch event pairs: eager steps right behind the replayed timed region supply them
        n_eager = min(args.steps, 20)
        trainer.use_graph = False
        trainer.step(inputs)                                # (one untimed eager step: allocator / MIOpen back in eager mode)
        ops.kernel_timer.enable(TIMED, reserve=96 * n_eager)
        for _ in range(n_eager):
            trainer.step(inputs)
        torch.cuda.synchronize()
        ops.kernel_timer.disable()
        timed_rows = ops.kernel_timer.rows()
        roofline_steps = float(n_eager)
        trainer.use_graph = True
    if not torch.isfinite(loss):
        raise SystemExit('non-finite loss in the timed region')

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    rank_spread = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # how the ranks differ: the slowest rank's host is what an eager job is paced by
        mine = torch.tensor([step_stats['step_ms']['median'], step_stats['step_ms']['max'], step_stats['host_enqueue_ms']['median'],
                             step_stats['host_enqueue_ms']['max'], step_stats['drain_ms']], device=dev, dtype=torch.float64)
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        names = ('step_ms_median', 'step_ms_max', 'host_enqueue_ms_median', 'host_enqueue_ms_max', 'drain_ms')
        rank_spread = {'max_over_ranks': {n: round(v, 3) for n, v in zip(names, hi.tolist())},
                       'min_over_ranks': {n: round(v, 3) for n, v in zip(names, lo.tolist())}}
    dt = t.item()

    # dense-contraction FLOPs of one step (forward + data-gradient + weight-gradient of every convolution),
    # counted by hooks over one extra untimed step; SURVEY 8d: 1.945 TFLOP at 832x256, B=8
    conv_flops = [0.0]
    import torch.nn.functional as F_
    real_conv2d = F_.conv2d

    def counting_conv2d(x, w, *a, **k):                  # every convolution of the model goes through F.conv2d (ConvLeaky, the
        out = real_conv2d(x, w, *a, **k)                 # cat-free decoder's raw contractions, the flow predictors)
        conv_flops[0] += 2.0 * out.numel() * w.shape[1] * w.shape[2] * w.shape[3] * (3.0 if x.requires_grad else 2.0)
        return out
    F_.conv2d = counting_conv2d
    was_graph, trainer.use_graph = trainer.use_graph, False          # (one eager step; a replay calls no Python)
    try:
        trainer.step(inputs)
        torch.cuda.synchronize()
    finally:
        F_.conv2d = real_conv2d
        trainer.use_graph = was_graph

    survey = None
    if rank == 0 and world == 1 and (graph_timing or not args.no_kernel_timing):
        # per-entry-point timings of ALL hand-written kernels: 3 extra (untimed) steps with an event pair attached to
        # the kernels of every C call
        was_graph, trainer.use_graph = trainer.use_graph, False
        ops.kernel_timer.enable(True)
        for _ in range(3):
            trainer.step(inputs)
        torch.cuda.synchronize()
        ops.kernel_timer.disable()
        trainer.use_graph = was_graph
        rows = ops.kernel_timer.rows()
        for r in rows:
            r.pop('total_us'); r.pop('total_bytes')
            if r['algorithmic_GBps'] is not None:
                r['frac_of_hbm_peak'] = round(r['algorithmic_GBps'] / HBM_PEAK_GBS, 3)
            r['launches_per_step'] = r.pop('launches') / 3.0
        survey = rows[:14]

    if rank == 0:
        roof = None
        if timed_rows:
            K = roofline_steps

            def entry(r):
                return {'entry': r['entry'], 'shape': r['shape'], 'avg_us': r['avg_us'], 'launches_per_step': r['launches'] / K,
                        'us_per_step': round(r['total_us'] / K, 2), 'algorithmic_bytes_per_launch': int(r['total_bytes'] / r['launches']),
                        'algorithmic_GBps': r['algorithmic_GBps'], 'frac': round((r['algorithmic_GBps'] or 0.0) / HBM_PEAK_GBS, 4)}
            loss_rows = [r for r in timed_rows if r['entry'] in LOSSES]
            timed_rows = [r for r in timed_rows if r['entry'] in CW]
            top = timed_rows[0]                      # rows() sorts by total time: the heaviest (entry point, shape)
            best = max(timed_rows, key=lambda r: r['algorithmic_GBps'] or 0.0)
            tot_us, tot_b = sum(r['total_us'] for r in timed_rows) / K, sum(r['total_bytes'] for r in timed_rows) / K
            traffic, traffic_note = measured_traffic(top['entry'], top['shape'])
            gbs = top['algorithmic_GBps'] or 0.0
            # `bound` names the roofline the fraction is taken against (HBM for every hand-written kernel); `limited_by` what the
            # counters say holds the kernel below it today
            limited = None
            if top['entry'] == 'unflow_corr_bwd':
                limited = ('valu: packed fp32 FMA issue (15.0 M VALU instructions in 51 M wave-cycles at level 2, LDS array busy 26 %, VALU issuing 41 % of all SIMD cycles, no bank '
                           'conflicts: profiles/r5_corr_bwd_mfma/pmc_sq_counters.txt; the matrix-core form of the same sums, csrc/corr_mfma.h, lost in the step at d = 4 '
                           '(81.2 vs 70.5 us) and is opt-in at d = 8: profiles/r5_corr_bwd_mfma.md)')
            # `bound` stays the contract's roofline ('hbm': `peak` and `frac` are HBM figures); `limiter` is the machine-readable name of what
            # the counters say actually holds this kernel ('valu' for the level-2 cost-volume backward), `limited_by` the evidence in words
            roof = {'bound': 'hbm', 'limiter': ('valu' if limited else None), 'limited_by': limited, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                    'traffic': traffic, 'traffic_source': traffic_note,
                    'kernel': '%s %s: the cost-volume / warp entry point with the largest time per step' % (top['entry'], top['shape']),
                    'measured_in': ('%d eager steps right behind the timed region (the timed region replays a hipGraph, which cannot carry per-launch events; --graph 0 times them inside it)' % int(roofline_steps)
                                    if graph_timing else 'the timed steps'),
                    'launches': top['launches'], 'avg_us': top['avg_us'],
                    'algorithmic_bytes_per_launch': int(top['total_bytes'] / top['launches']),
                    'aggregate': {'what': 'every cost-volume and warp launch of a step (all pyramid levels, forward and backward)',
                                  'algorithmic_bytes_per_step': int(tot_b), 'us_per_step': round(tot_us, 1),
                                  'achieved': round(tot_b / tot_us / 1e3, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                  'frac': round(tot_b / tot_us / 1e3 / HBM_PEAK_GBS, 4),
                                  'launches_per_step': sum(r['launches'] for r in timed_rows) / K,
                                  'per_level': [entry(r) for r in timed_rows]},
                    'best': entry(best)}
            if loss_rows:
                l_us, l_b = sum(r['total_us'] for r in loss_rows) / K, sum(r['total_bytes'] for r in loss_rows) / K
                roof['losses'] = {'what': 'every occlusion-weight / masked-L1 / SSIM / smoothness / consistency launch of a step (three scales, both '
                                          'directions batched as 2B), forward and backward; bound: hbm',
                                  'algorithmic_bytes_per_step': int(l_b), 'us_per_step': round(l_us, 1),
                                  'achieved': round(l_b / l_us / 1e3, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                  'frac': round(l_b / l_us / 1e3 / HBM_PEAK_GBS, 4),
                                  'launches_per_step': sum(r['launches'] for r in loss_rows) / K,
                                  'per_entry': [entry(r) for r in loss_rows]}
        base = None
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(args.cpu_sample)
        elif world > 1:
            base = 'see the N=1 line (the CPU oracle is timed on rank 0 at N=1 only)'
        pairs = 2 * args.batch * world * args.steps
        out = {
            'metric': 'frame-pairs/s (train step) at %dx%d bs=%d' % (fw, fh, args.batch),
            'value': round(pairs / dt, 2), 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'contended_host': bool(args.contended_host), 'dtype': 'f32' if args.precision == 'fp32' else 'bf16', 'data': 'synthetic',
            'step_mode': (('hipGraph replay: graph(forward + backward + gradient pack), one all-reduce of the flat gradient, graph(Adam)'
                           if (world > 1 or args.force_ddp) else 'hipGraph replay') if args.graph else
                          ('eager, gradient pieces all-reduced from hooks during backward' if (world > 1 or args.force_ddp) else 'eager')),
            'config': {'workload': '%dx%d triplets, bs=%d per GPU, %s, corr d=4 + warp + occlusion losses, '
                                   'fwd+bwd+Adam (%s)' % (fw, fh, args.batch, args.precision,
                                                          'BASELINE configs[%d]' % (1 if args.precision == 'fp32' else 2) if (fh, fw, args.batch) == (H, W, B_PER_GPU)
                                                          else 'BASELINE configs[3]' if (fh, fw, args.batch) == (448, 1024, 4) else 'not a BASELINE configuration'),
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                       'conv_memory_format': 'channels_last' if cfg.channels_last else 'NCHW',
                       'triplets_per_s': round(pairs / 2 / dt, 2),
                       'loss_launch_form': 'one per loss over the scales' if args.multiscale_losses else 'one per loss and scale'},
            'step_ms': step_stats['step_ms'], 'host_enqueue_ms': step_stats['host_enqueue_ms'], 'drain_ms': step_stats['drain_ms'],
            'pairs_per_s_at_median_step': round(2 * args.batch * world / (step_stats['step_ms']['median'] * 1e-3), 2),
            'all_step_ms': step_stats['all_step_ms'], 'host_gc': step_stats['host_gc'], 'rank_spread': rank_spread,
            'roofline': roof, 'cpu_baseline': base,
            # whole-step lower bound on the conv stacks' MFMA utilisation: conv FLOPs / (entire step time);
            # profiles/ holds the per-kernel split (convolutions alone: see DESIGN.md section 4)
            'conv_stack': None if not conv_flops[0] else {
                'bound': 'mfma', 'flops_per_step': conv_flops[0],
                'achieved_lower_bound': round(conv_flops[0] / (dt / args.steps) / 1e12, 1),
                'peak': MFMA_PEAK_TFLOPS[args.precision], 'unit': 'TFLOP/s',
                'frac_lower_bound': round(conv_flops[0] / (dt / args.steps) / 1e12 / MFMA_PEAK_TFLOPS[args.precision], 4)},
            'kernel_survey': survey,
        }
        # MFMA utilisation of the convolution kernels alone, from hardware counters (a committed rocprofv3 --pmc capture of
        # this very command: tools/gpu_r3_profile.sh -> tools/summarize_mfma.py); the live figure above divides by the whole step
        mf = latest_profile('conv_mfma_%s.json' % args.precision)
        if out['conv_stack'] is not None and mf is not None and (fh, fw, args.batch) == (H, W, B_PER_GPU):
            mfd = json.load(open(mf))
            ck = mfd['conv_kernels_only']
            # (the counters cover MIOpen's / CK's convolution kernels only: what decides them is WHICH convolutions the step runs, i.e. the
            # package's Python sources, not the hand-written kernel sources)
            same = mfd.get('host_sources_sha16') == host_sources_sha16()
            out['conv_stack']['conv_kernels_only'] = {'achieved': ck['achieved_tflops'], 'frac': ck['frac_of_peak'], 'ms_per_step': ck['ms_per_step'],
                                                      'mfma_tflop_per_step_counted': ck['mfma_tflop_per_step'],
                                                      'measured_on_these_sources': same,      # False: a capture of an earlier source state, kept for reference
                                                      'source': 'profiles/%s (SQ_INSTS_VALU_MFMA_MOPS_* x 512 / kernel time, separate --pmc and --kernel-trace runs; NOT measured in this run)' % os.path.basename(mf)}
        line = json.dumps(out)
    else:
        line = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes a version banner to the C stdout stream (fully buffered when stdout is a file or a pipe: it would land behind
        # the JSON at exit); drain it first so that the JSON is the LAST line of stdout, after the process group is gone
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)


if __name__ == '__main__':
    main()
