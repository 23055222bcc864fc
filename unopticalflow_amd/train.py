"""Drop-in for the reference's ``train.py --mode flow`` (train.py:33-226) on MI355X.

    python -m unopticalflow_amd.train -c unopticalflow_amd/config/kitti.yaml --gpu 0 --mode flow \
        --prepared_save_dir data_s1 --model_dir models [--synthetic]
    python -m unopticalflow_amd.train -c ... --gpu 0,1,2,3,4,5,6,7 --multi_gpu          # starts its 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m unopticalflow_amd.train -c ... --gpu 0,1,2,3,4,5,6,7 --multi_gpu              # (or behind a launcher)

Same flags and yaml keys as the reference.  ``--multi_gpu`` keeps its meaning -- the global batch is
batch_size x num_gpus and num_iterations / num_gpus (train.py:211-213) -- but runs one process per
GPU with gradients all-reduced over RCCL instead of single-process DataParallel.  Checkpoints keep
the reference's dict (train.py:23-24) with unwrapped keys.
"""
import argparse
import os
import pickle
import shutil
import time

import torch
import yaml

from .core.networks import get_model
from .data import DecodedTriplets, DeviceTripletLoader, PreparedTriplets, SyntheticTriplets
from .parallel import init_distributed
from .trainer import FlowTrainer


def print_loss(loss_pack, iter_=None, extra=''):
    """Same line as the reference's Visualizer.print_loss (core/visualize/visualizer.py:32-48)."""
    v = {k: loss_pack[k].mean().detach().cpu().numpy() for k in loss_pack}
    print('iter: {4}, loss_pixel: {0:.6f}, loss_ssim: {1:.6f}, loss_flow_smooth: {2:.6f}, loss_flow_consis: {3:.6f}'
          .format(v['loss_pixel'], v['loss_ssim'], v['loss_flow_smooth'], v['loss_flow_consis'], iter_) + extra, flush=True)


def gpu_ids(gpu_arg):
    """``--gpu 4,5,6,7`` -> [4, 5, 6, 7] (reference train.py:198 sets CUDA_VISIBLE_DEVICES from the same flag).  With
    one process per GPU, rank r drives the r-th id of the list (relative to whatever HIP_VISIBLE_DEVICES already allows)."""
    ids = [int(x) for x in str(gpu_arg).split(',') if x.strip() != '']
    if not ids:
        raise ValueError('--gpu needs at least one device id')
    return ids


def training_items(cfg, world):
    """Number of dataset items and the per-rank batch so that EVERY rank's loader yields exactly
    ``num_iterations - iter_start`` batches: the dataset is sized with the GLOBAL batch (reference train.py:110:
    ``(num_iterations - iter_start) * batch_size``), and DistributedSampler hands each rank 1/world of it."""
    per_rank = cfg.batch_size // world
    if per_rank * world != cfg.batch_size:
        raise ValueError('batch size {} is not divisible by {} ranks'.format(cfg.batch_size, world))
    return (cfg.num_iterations - cfg.iter_start) * per_rank * world, per_rank


def evaluate_during_training(cfg, model, iter_):
    """Periodic KITTI evaluation of the reference loop (train.py:157-162: test_kitti_2012 / test_kitti_2015 every
    ``--test_interval`` iterations unless ``--no_test``), results appended to ``log.pkl``.  Runs only when the ground-truth
    directories of the yaml exist (they do not on a synthetic run); otherwise it says so once and is skipped."""
    from . import test as test_cli
    dirs = [(name, getattr(cfg, key, None)) for name, key in (('kitti_2012', 'gt_2012_dir'), ('kitti_2015', 'gt_2015_dir'))]
    dirs = [(n, d) for n, d in dirs if d and os.path.isdir(d)]
    if not dirs:
        if not getattr(cfg, '_warned_no_gt', False):
            print('evaluation skipped: no ground-truth directory (gt_2012_dir / gt_2015_dir) on this machine; '
                  'pass --no_test to silence this', flush=True)
            cfg._warned_no_gt = True
        return None
    from . import evaluation as ev
    cache = cfg.__dict__.setdefault('_gt_cache', {})                 # ground truth is read once (train.py:113-117)
    was_training = model.training
    model.eval()
    results = {}
    try:
        with torch.no_grad():
            for name, d in dirs:
                if name not in cache:
                    gt_flows, noc_masks = ev.load_gt_flow_kitti(d, name)
                    cache[name] = (gt_flows, noc_masks, ev.load_gt_mask(d) if name == 'kitti_2015' else None)
                gt_flows, noc_masks, gt_masks = cache[name]
                if name == 'kitti_2012':
                    results[name] = test_cli.test_kitti_2012(cfg, model, gt_flows, noc_masks)
                else:
                    results[name] = test_cli.test_kitti_2015(cfg, model, gt_flows, noc_masks, gt_masks)
    finally:
        model.train(was_training)
    log = []
    if os.path.exists(cfg.log_dump_dir):
        with open(cfg.log_dump_dir, 'rb') as f:
            log = pickle.load(f)
    log.append({'iteration': iter_, **results})
    with open(cfg.log_dump_dir, 'wb') as f:
        pickle.dump(log, f)
    return results


def make_eval_group(timeout_s=4 * 3600):
    """A gloo group with its own (long) timeout for the barrier behind rank 0's periodic evaluation: the RCCL group's
    watchdog (10 min) would kill the waiting ranks.  Created ONCE at start-up, while every rank is present -- group
    construction is itself a rendezvous, and doing it lazily would have ranks 1..N enter it minutes before rank 0."""
    import datetime
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    return dist.new_group(backend='gloo', timeout=datetime.timedelta(seconds=timeout_s))


def eval_barrier(group):
    """All ranks meet after rank 0's periodic evaluation (``group`` from ``make_eval_group``; None: one process)."""
    import torch.distributed as dist
    if group is None:
        return
    torch.cuda.synchronize()
    dist.barrier(group=group)


def train(cfg):
    local = int(os.environ.get('LOCAL_RANK', '0'))
    ids = gpu_ids(getattr(cfg, 'gpu', '0'))
    if local >= len(ids):
        raise ValueError('LOCAL_RANK {} but --gpu lists only {} device(s)'.format(local, len(ids)))
    if not torch.cuda.is_available():
        raise RuntimeError('unopticalflow_amd.train needs an MI355X; there is no CPU path')
    if ids[local] >= torch.cuda.device_count():
        raise ValueError('--gpu {}: this process sees {} device(s)'.format(ids[local], torch.cuda.device_count()))
    rank, local_rank, world = init_distributed('nccl', device_index=ids[local])
    torch.cuda.set_device(ids[local])
    dev = torch.device('cuda', ids[local])
    eval_group = make_eval_group() if not getattr(cfg, 'no_test', False) else None

    from . import tuning
    if getattr(cfg, 'miopen_find', 1):
        tuning.enable_miopen_tuning()
        if rank == 0 and torch.backends.cudnn.benchmark and tuple(cfg.img_hw) not in ((256, 832), (448, 1024)):
            print('MIOpen find mode: the shipped find-db covers 832x256 (bs 8/GPU) and 1024x448 (bs 4/GPU); other shapes are '
                  'measured once during the first iterations (20+ minutes for a full-size shape) -- pass --miopen_find 0 to skip that.', flush=True)
    if getattr(cfg, 'channels_last', None) is None:
        cfg.channels_last = tuning.default_channels_last()              # NHWC only with MIOpen's measured picks
    model = get_model(cfg.mode)(cfg).to(dev)
    trainer = FlowTrainer(cfg, model, distributed=(world > 1), use_graph=bool(getattr(cfg, 'graph', 1)), gc_freeze_after=2)
    if cfg.resume:                                                     # train.py:42-46
        name = 'iter_{}.pth'.format(cfg.iter_start) if cfg.iter_start > 0 else 'last.pth'
        cfg.iter_start = trainer.load(os.path.join(cfg.model_dir, name), map_location=dev)
    elif cfg.flow_pretrained_model:                                    # train.py:47-61 (flow-mode keys are bare)
        data = torch.load(cfg.flow_pretrained_model, map_location=dev)['model_state_dict']
        data = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in data.items()}
        missing, unexpected = model.load_state_dict(data, strict=False)
        print(missing); print(unexpected)
        print('Load Flow Pretrained Model from ' + cfg.flow_pretrained_model)

    n_items, per_rank = training_items(cfg, world)
    if cfg.synthetic:
        batches = SyntheticTriplets(per_rank, cfg.img_hw, dev, seed=rank)
    else:
        data_dir = os.path.join(cfg.prepared_base_dir, cfg.prepared_save_dir)
        if not os.path.exists(os.path.join(data_dir, 'train.txt')):
            raise FileNotFoundError('no prepared triplets under {} (raw-dataset preparation is outside this '
                                    'package: run the reference\'s prepare_data_mp once, or use --synthetic)'.format(data_dir))
        ds = (PreparedTriplets if cfg.host_input else DecodedTriplets)(data_dir, cfg.num_scales, cfg.img_hw, n_items)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True) if world > 1 else None
        if cfg.host_input:                                             # the reference's all-CPU pipeline (PIL decode, OpenCV's 8-bit resize restated: data.resize_linear_u8)
            loader = torch.utils.data.DataLoader(ds, batch_size=per_rank, shuffle=(sampler is None), sampler=sampler,
                                                 num_workers=cfg.num_workers, drop_last=False, pin_memory=True)
            batches = (b.to(dev, non_blocking=True) for b in loader)
        else:                                                          # decode on the CPU, the rest on the GPU
            batches = DeviceTripletLoader(ds, per_rank, dev, cfg.img_hw, cfg.num_workers, sampler)

    if rank == 0:
        print('starting iteration: {}.'.format(cfg.iter_start))
    t_last, n_last = time.perf_counter(), 0
    for iter_, inputs in enumerate(batches):
        iter_ = iter_ + cfg.iter_start
        if iter_ >= cfg.num_iterations:
            break
        loss, loss_pack = trainer.step(inputs)                         # train.py:137-152
        n_last += 1
        if rank == 0 and iter_ % cfg.log_interval == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t_last
            rate = n_last * cfg.batch_size / dt
            print_loss(loss_pack, iter_=iter_, extra=', samples/s: {:.1f}, pairs/s: {:.1f}'.format(rate, 2 * rate))
            t_last, n_last = time.perf_counter(), 0
        if not cfg.no_test and (iter_ + 1) % cfg.test_interval == 0:       # train.py:157-162
            # rank 0 evaluates (batch-1 inference shapes are not in the find-db: heuristics, no exhaustive find); the other
            # ranks wait HERE in a barrier with a long timeout instead of in the next step's all-reduce, whose watchdog
            # (10 min) would abort the job during a ~400-image evaluation
            bench_mode = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = False
            try:
                if rank == 0:
                    evaluate_during_training(cfg, model, iter_)
            finally:
                torch.backends.cudnn.benchmark = bench_mode
                eval_barrier(eval_group)
        if rank == 0 and (iter_ + 1) % cfg.save_interval == 0:         # train.py:153-155
            trainer.iteration = iter_
            trainer.save(os.path.join(cfg.model_dir, 'iter_{}.pth'.format(iter_)))
            trainer.save(os.path.join(cfg.model_dir, 'last.pth'))
    return trainer


def main(argv=None):
    ap = argparse.ArgumentParser(description='UnOpticalFlow flow-stage training on MI355X.')
    ap.add_argument('-c', '--config_file', default=None, help='config file.')
    ap.add_argument('-g', '--gpu', type=str, default='0', help='gpu id.')
    ap.add_argument('--batch_size', type=int, default=8, help='batch size.')
    ap.add_argument('--iter_start', type=int, default=0, help='starting iteration.')
    ap.add_argument('--lr', type=float, default=0.0001, help='learning rate')
    ap.add_argument('--num_workers', type=int, default=4, help='number of workers.')
    ap.add_argument('--log_interval', type=int, default=100, help='interval for printing loss.')
    ap.add_argument('--test_interval', type=int, default=2000, help='interval for evaluation.')
    ap.add_argument('--save_interval', type=int, default=2000, help='interval for saving models.')
    ap.add_argument('--mode', type=str, default='flow', help='training mode.')
    ap.add_argument('--model_dir', type=str, default=None, help='directory for saving models')
    ap.add_argument('--prepared_save_dir', type=str, default='data_s1', help='directory name for generated training dataset')
    ap.add_argument('--flow_pretrained_model', type=str, default=None, help='directory for loading flow pretrained models')
    ap.add_argument('--depth_pretrained_model', type=str, default=None, help='(unsupported: depth stage is out of scope)')
    ap.add_argument('--resume', action='store_true', help='to resume training.')
    ap.add_argument('--multi_gpu', action='store_true', help='to use multiple gpu for training.')
    ap.add_argument('--no_test', action='store_true', help='without evaluation.')
    # additions
    ap.add_argument('--synthetic', action='store_true', help='train on on-device synthetic triplets (no dataset).')
    ap.add_argument('--align_corners', type=int, default=0, help='grid_sample generation: 0 torch>=1.3, 1 torch 1.2.0.')
    ap.add_argument('--num_iterations', type=int, default=None, help='override the yaml value.')
    ap.add_argument('--precision', type=str, default='fp32', choices=['fp32', 'bf16'], help='conv-stack precision.')
    ap.add_argument('--channels_last', type=int, default=None, help='memory format of the conv stacks (1 NHWC, 0 NCHW); default: 1 when the shipped MIOpen find-db is in use (same device and MIOpen build), else 0.')
    ap.add_argument('--miopen_find', type=int, default=1, help='1: use the shipped MIOpen find-db + benchmark mode.')
    ap.add_argument('--graph', type=int, default=1, help='1 (default): replay the train step as a hipGraph -- one process: forward + backward + Adam in one graph; '
                    'several ranks: forward + backward, one all-reduce, Adam graph (the host enqueues 4 things per step instead of ~3000); '
                    '0: eager launches, gradient pieces all-reduced from hooks during backward.')
    ap.add_argument('--host_input', type=int, default=0, help='1: resize/flip/scale on the CPU workers (PIL) instead of the GPU.')
    args = ap.parse_args(argv)
    if args.config_file is None:
        raise ValueError('config file needed. -c --config_file.')
    if not os.path.exists(args.config_file):
        raise ValueError('config file not found.')
    if args.depth_pretrained_model:
        raise ValueError('--depth_pretrained_model belongs to the depth stage, which this package does not cover')
    num_gpus = len(str(args.gpu).split(','))
    if (args.multi_gpu and num_gpus <= 1) or ((not args.multi_gpu) and num_gpus > 1):
        raise ValueError('Error! the number of gpus used in the --gpu argument does not match the argument --multi_gpu.')
    from .launch import launched_by_torchrun, spawn_ranks
    if args.multi_gpu and not launched_by_torchrun():
        # the reference's one-command form (train.py:208-214): start one process per listed GPU ourselves.  This parent has
        # imported torch but made no GPU call; the children are fresh interpreters, rank r drives the r-th id of --gpu.
        import sys
        cmd = [sys.executable, '-m', 'unopticalflow_amd.train'] + (list(argv) if argv is not None else sys.argv[1:])
        raise SystemExit(spawn_ranks(cmd, num_gpus))
    if args.model_dir is None:
        args.model_dir = os.path.join('models', os.path.splitext(os.path.split(args.config_file)[1])[0])
    args.model_dir = os.path.join(os.getcwd(), args.model_dir, args.mode)
    os.makedirs(args.model_dir, exist_ok=True)
    with open(args.config_file, 'r') as f:
        cfg = yaml.safe_load(f)
    cfg['img_hw'] = (cfg['img_hw'][0], cfg['img_hw'][1])
    cfg['log_dump_dir'] = os.path.join(args.model_dir, 'log.pkl')
    if int(os.environ.get('RANK', '0')) == 0:
        shutil.copy(args.config_file, args.model_dir)
    num_iter_override = args.num_iterations
    for attr, val in vars(args).items():                               # CLI attrs overwrite yaml keys (train.py:203-205)
        if attr == 'num_iterations' and val is None:
            continue
        cfg[attr] = val
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.multi_gpu:
        if world != num_gpus:
            raise ValueError('--multi_gpu with {} gpus but the launcher started {} process(es): drop the launcher (this '
                             'command starts its own ranks) or use --nproc-per-node {}'.format(num_gpus, world, num_gpus))
        cfg['batch_size'] = cfg['batch_size'] * num_gpus               # train.py:211-213
        if num_iter_override is None:
            cfg['num_iterations'] = int(cfg['num_iterations'] / num_gpus)

    class pObject(object):
        pass
    cfg_new = pObject()
    for k, v in cfg.items():
        setattr(cfg_new, k, v)
    if int(os.environ.get('RANK', '0')) == 0:
        with open(os.path.join(args.model_dir, 'config.pkl'), 'wb') as f:
            pickle.dump(cfg, f)
    return train(cfg_new)


if __name__ == '__main__':
    main()
