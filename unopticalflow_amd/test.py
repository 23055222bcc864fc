"""Drop-in for the flow part of the reference's ``test.py`` (test.py:16-76,200-260).

    python -m unopticalflow_amd.test -c cfg.yaml --gpu 0 --mode flow --task synthetic_flow \
        --pretrained_model models/.../last.pth

``--task kitti_flow`` evaluates on KITTI 2015 (``gt_2015_dir`` of the yaml, or ``--gt_dir``) exactly as
the reference's test_kitti_2015 (test.py:43-76): per pair ``inference_flow`` at ``img_hw``, then
``eval_flow_avg`` (EPE / noc / occ / Fl, moving-object split) from ``unopticalflow_amd.evaluation``.
``--task kitti_flow_2012`` is test_kitti_2012 (test.py:16-41).  ``--task synthetic_flow`` needs no
dataset: synthetic pairs with a known translation, EPE with the reference's formula
(evaluate_flow.py:131-134).
"""
import argparse
import os

import torch
import yaml

from .core.networks import Model_flow
from . import evaluation as ev


def epe(flow, gt):
    d = flow - gt
    return torch.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2).mean()


def _predict(cfg, model, dataset):
    flows = []
    dev = next(model.parameters()).device
    for idx in range(len(dataset)):
        img = dataset[idx][None].to(dev)
        img_h = int(img.shape[2] / 2)
        with torch.no_grad():
            flow = model.inference_flow(img[:, :, :img_h, :], img[:, :, img_h:, :])
        flows.append(flow[0].detach().cpu().numpy().transpose(1, 2, 0))
    return flows


def test_kitti_2012(cfg, model, gt_flows, noc_masks, num=None):
    """reference test.py:16-41"""
    res = ev.eval_flow_avg(gt_flows, noc_masks, _predict(cfg, model, ev.KITTI_2012(cfg.gt_2012_dir, cfg.img_hw, num)), cfg)
    print('CONFIG: {0}, mode: {1}'.format(getattr(cfg, 'config_file', None), cfg.mode))
    print('[EVAL] [KITTI 2012]')
    print(res)
    return res


def test_kitti_2015(cfg, model, gt_flows, noc_masks, gt_masks, depth_save_dir=None, num=None):
    """reference test.py:43-76"""
    res = ev.eval_flow_avg(gt_flows, noc_masks, _predict(cfg, model, ev.KITTI_2015(cfg.gt_2015_dir, cfg.img_hw, num)), cfg,
                           moving_masks=gt_masks)
    print('CONFIG: {0}, mode: {1}'.format(getattr(cfg, 'config_file', None), cfg.mode))
    print('[EVAL] [KITTI 2015]')
    print(res)
    return res


def test_synthetic_flow(cfg, model, n=8, shift=(3.0, 1.0)):
    dev = next(model.parameters()).device
    H, W = cfg.img_hw
    g = torch.Generator(device=dev); g.manual_seed(0)
    base = torch.rand((n, 3, H + 16, W + 16), generator=g, device=dev)
    base = torch.nn.functional.avg_pool2d(base, 5, 1, 2)              # smooth texture
    sx, sy = int(shift[0]), int(shift[1])
    img1 = base[:, :, 8:8 + H, 8:8 + W].contiguous()
    img2 = base[:, :, 8 - sy:8 - sy + H, 8 - sx:8 - sx + W].contiguous()
    with torch.no_grad():
        flow = model.inference_flow(img1, img2)
    gt = torch.zeros_like(flow); gt[:, 0] = shift[0]; gt[:, 1] = shift[1]
    res = {'epe': float(epe(flow, gt)), 'pairs': n}
    print('[EVAL] [synthetic translation {}]'.format(shift)); print(res)
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description='UnOpticalFlow flow-stage testing on MI355X.')
    ap.add_argument('-c', '--config_file', default=None, help='config file.')
    ap.add_argument('-g', '--gpu', type=str, default='0', help='gpu id.')
    ap.add_argument('--mode', type=str, default='flow', help='mode for testing.')
    ap.add_argument('--task', type=str, default='synthetic_flow', help='kitti_flow or synthetic_flow')
    ap.add_argument('--pretrained_model', type=str, default=None, help='directory for loading flow pretrained models')
    ap.add_argument('--result_dir', type=str, default=None, help='directory for saving predictions')
    ap.add_argument('--align_corners', type=int, default=0)
    ap.add_argument('--gt_dir', type=str, default=None, help='KITTI training dir (overrides gt_2015_dir / gt_2012_dir of the yaml)')
    ap.add_argument('--num_eval', type=int, default=None, help='evaluate only the first N pairs')
    args = ap.parse_args(argv)
    if not os.path.exists(args.config_file):
        raise ValueError('config file not found.')
    with open(args.config_file, 'r') as f:
        cfg = yaml.safe_load(f)
    cfg['img_hw'] = (cfg['img_hw'][0], cfg['img_hw'][1])
    cfg['model_dir'] = args.result_dir
    for attr, val in vars(args).items():
        cfg[attr] = val

    class pObject(object):
        pass
    cfg_new = pObject()
    for k, v in cfg.items():
        setattr(cfg_new, k, v)
    if args.mode != 'flow':
        raise ValueError('only --mode flow is covered by this package')
    # --gpu selects the device like the reference's CUDA_VISIBLE_DEVICES=args.gpu (test.py:213); the first id is used
    gpu = int(str(args.gpu).split(',')[0])
    if not torch.cuda.is_available():
        raise RuntimeError('unopticalflow_amd.test needs an MI355X; there is no CPU path')
    if gpu >= torch.cuda.device_count():
        raise ValueError('--gpu {}: this process sees {} device(s)'.format(gpu, torch.cuda.device_count()))
    dev = torch.device('cuda', gpu)
    torch.cuda.set_device(dev)
    model = Model_flow(cfg_new).to(dev)
    if args.pretrained_model:
        weights = torch.load(args.pretrained_model, map_location=dev)
        sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in weights['model_state_dict'].items()}
        model.load_state_dict(sd)
    model.eval()
    print('Model Loaded.')
    if args.task == 'synthetic_flow':
        return test_synthetic_flow(cfg_new, model)
    if args.task == 'kitti_flow':
        if args.gt_dir:
            cfg_new.gt_2015_dir = args.gt_dir
        gt_flows, noc_masks = ev.load_gt_flow_kitti(cfg_new.gt_2015_dir, 'kitti_2015', args.num_eval)
        gt_masks = ev.load_gt_mask(cfg_new.gt_2015_dir, args.num_eval or 200)
        return test_kitti_2015(cfg_new, model, gt_flows, noc_masks, gt_masks, num=args.num_eval)
    if args.task == 'kitti_flow_2012':
        if args.gt_dir:
            cfg_new.gt_2012_dir = args.gt_dir
        gt_flows, noc_masks = ev.load_gt_flow_kitti(cfg_new.gt_2012_dir, 'kitti_2012', args.num_eval)
        return test_kitti_2012(cfg_new, model, gt_flows, noc_masks, num=args.num_eval)
    raise ValueError('unknown task {}'.format(args.task))


if __name__ == '__main__':
    main()
