"""Where a replayed multi-rank step spends its time when the ranks SHARE one GPU (the one-GPU rehearsal of bench.py,
UNFLOW_BENCH_ONE_GPU=1: gloo, all ranks on device 0).  The rehearsal shows single steps of 10-25 s between 60 ms ones; this probe
splits every step into its phases (input copy + graph A, the flat all-reduce, graph B) with a device synchronisation and a host
stamp after each, per rank, so the stall can be put on the GPU sharing, on gloo's staged all-reduce or on the graph replay.

    python3 tools/probes/two_rank_phases.py [--ranks 2] [--steps 16] [--mode graph|eager|graph_noexchange]

Prints one JSON object per rank (rank 0 on stdout, the others on stderr).  Numbers are diagnostic only.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ranks', type=int, default=2)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--mode', default='graph', choices=('graph', 'eager', 'graph_noexchange'))
    ap.add_argument('--nosync', action='store_true', help='host stamps only: no device synchronisation and no barrier inside the loop '
                                                          '(what a training loop does); the phase columns are then host enqueue times')
    return ap.parse_args()


def main():
    args = parse()
    from unopticalflow_amd import launch
    if not launch.launched_by_torchrun():
        if args.ranks > 1:
            raise SystemExit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.ranks))
        os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(launch.free_port()))
    import types
    import torch
    import torch.distributed as dist
    from unopticalflow_amd import get_model, _lib, tuning
    from unopticalflow_amd.parallel import init_distributed
    from unopticalflow_amd.trainer import FlowTrainer

    _lib.load()
    rank, _, world = init_distributed('gloo', device_index=0, force=True)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    tuning.enable_miopen_tuning()
    cfg = types.SimpleNamespace(mode='flow', dataset='kitti_depth', num_scales=3, h_flow_consist_alpha=3.0, h_flow_consist_beta=0.05,
                                w_ssim=0.85, w_flow_smooth=10.0, w_flow_consis=0.01, lr=1e-4, align_corners=False, precision='fp32',
                                weight_shadows=False, fused_warp_corr=False, channels_last=tuning.default_channels_last())
    torch.manual_seed(1234)
    model = get_model('flow')(cfg).to(dev)
    graph = args.mode != 'eager'
    trainer = FlowTrainer(cfg, model, distributed=True, use_graph=graph, single_rank_collectives=(world == 1), gc_freeze_after=None)
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    inputs = torch.rand((8, 3, 3 * 256, 832), generator=gen, device=dev)

    def stamp(force=False):
        if force or not args.nosync:
            torch.cuda.synchronize()
        return time.perf_counter()

    t = stamp(True)
    trainer.step(inputs)                          # capture (graph modes) / first eager step
    first = stamp(True) - t
    rows = []
    for _ in range(args.steps):
        if not args.nosync:
            dist.barrier()
        t0 = stamp()
        if graph:
            trainer._static_in.copy_(inputs)
            trainer._graph.replay()
            t1 = stamp()
            if args.mode == 'graph':
                trainer.grads.all_reduce_flat()
            t2 = stamp()
            trainer._graph_opt.replay()
            t3 = stamp()
        else:
            trainer.step(inputs)
            t1 = t2 = t3 = stamp()
        rows.append([round((b - a) * 1e3, 2) for a, b in ((t0, t1), (t1, t2), (t2, t3))])
    stamp(True)
    dist.barrier()
    tot = [sum(r) for r in rows]
    out = {'rank': rank, 'world': world, 'mode': args.mode, 'nosync': bool(args.nosync), 'first_step_ms': round(first * 1e3, 1),
           'phases': 'fwd+bwd(+pack) | all-reduce | adam' if graph else 'whole eager step | - | -',
           'median_ms': [sorted(c)[len(c) // 2] for c in zip(*rows)], 'max_ms': [max(c) for c in zip(*rows)],
           'step_total_ms': [round(x, 1) for x in tot], 'rows_ms': rows}
    (sys.stdout if rank == 0 else sys.stderr).write(json.dumps(out) + '\n')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
