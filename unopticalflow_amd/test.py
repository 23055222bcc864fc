"""Drop-in for the flow part of the reference's ``test.py`` (test.py:16-76,200-260).

    python -m unopticalflow_amd.test -c cfg.yaml --gpu 0 --mode flow --task synthetic_flow \
        --pretrained_model models/.../last.pth

``--task kitti_flow`` needs the KITTI 2012/2015 flow PNGs and their decoder, the evaluation stage
listed as the next row (N1) in SURVEY.md section 8f; it is not implemented in this round and says
so.  ``--task synthetic_flow`` runs ``inference_flow`` on synthetic pairs with a known translation
and reports the end-point error with the reference's EPE formula (evaluate_flow.py:131-134).
"""
import argparse
import os

import torch
import yaml

from .core.networks import Model_flow


def epe(flow, gt):
    d = flow - gt
    return torch.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2).mean()


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
    model = Model_flow(cfg_new).cuda()
    if args.pretrained_model:
        weights = torch.load(args.pretrained_model, map_location='cuda')
        sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in weights['model_state_dict'].items()}
        model.load_state_dict(sd)
    model.eval()
    print('Model Loaded.')
    if args.task == 'synthetic_flow':
        return test_synthetic_flow(cfg_new, model)
    if args.task == 'kitti_flow':
        raise NotImplementedError('KITTI flow evaluation (dataset loader + flow-PNG decode + eval_flow_avg) is the '
                                  'next row after the hot path (SURVEY.md 8f N1); not part of this round')
    raise ValueError('unknown task {}'.format(args.task))


if __name__ == '__main__':
    main()
